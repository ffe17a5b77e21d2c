O=gpurun_out/exp_n20; mkdir -p $O
run() { n=$1; shift; timeout -k 10 200 python bench.py "$@" --no-cpu-baseline --no-target-leg > $O/$n.json 2> $O/$n.err || { echo "FAILED $n"; exit 1; }; python - $O/$n.json $n <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
c = d.get("closed_loop", {})
print("%-28s %8.3f M/s %8.4f ms  p50 %.4f  iters %.1f/%d  failed/step mean %s max %s itmax %s" % (sys.argv[2], d["value"]/1e6, d["ms_per_step"], d["p50_batch_ms"], d["iters_mean"], d["iters_max"],
      c.get("failed_frac_per_step", {}).get("mean"), c.get("failed_frac_per_step", {}).get("max"), c.get("iters_max")))
P
}
R="--workload rollout_6dof_B8192_N20 --steps 500"
S="--workload mpc_6dof_B4096_N20 --steps 200 --pipeline 1"
run roll_none $R
run roll_100_200 $R --phase-iters 100,200
run roll_100 $R --phase-iters 100,-1 --phase-scale "64,1,1,1;0,0,0,0"
run roll_200 $R --phase-iters 200,-1 --phase-scale "64,1,1,1;0,0,0,0"
run ser_none $S
run ser_100_200 $S --phase-iters 100,200
run ser_200 $S --phase-iters 200,-1 --phase-scale "64,1,1,1;0,0,0,0"
