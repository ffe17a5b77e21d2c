# The bench lines of a round, one JSON file each (run on the GPU box from the repository root).
ROUND=${ROUND:-r03}; O=gpurun_out/bench_$ROUND; mkdir -p $O
run() { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "FAILED $n"; exit 1; }; python - $O/$n.json $n <<'P'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("%-34s %10.3f M/s  %8.3f ms/batch  p50 %7.3f ms  iters %.1f/%d  frac %.3f  solved %.5f  inflight %d" % (
    sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d["p50_batch_ms"], d["iters_mean"], d["iters_max"], r["frac"],
    d["solved_frac"], d["config"]["batches_in_flight"]))
if "closed_loop" in d: print("    closed loop:", json.dumps(d["closed_loop"])[:400])
if "cpu_baseline" in d: print("    cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("single_core", {}).get("value"))
P
}
run n50_default --steps 20 --warmup 3
run n50_gridcap_off --steps 10 --no-cpu-baseline --grid-cap -1 --no-target-leg
run n50_dense --steps 10 --no-cpu-baseline --lowrank -1 --no-target-leg
run n50_B32768 --workload mpc_6dof_B32768_N50 --steps 40 --no-cpu-baseline --no-target-leg
run n50_B32768_serial --workload mpc_6dof_B32768_N50 --steps 40 --pipeline 1 --no-cpu-baseline --no-target-leg
run n50_two_in_flight --steps 10 --pipeline 2 --no-cpu-baseline --no-target-leg
run n20_6dof --workload mpc_6dof_B4096_N20 --steps 200
run n20_6dof_serial --workload mpc_6dof_B4096_N20 --steps 200 --pipeline 1 --no-cpu-baseline --no-target-leg
run n20_6dof_fine5 --workload mpc_6dof_B4096_N20 --steps 200 --pipeline 1 --check-fine 5 --no-cpu-baseline --no-target-leg
run n20_6dof_stress --workload mpc_6dof_B4096_N20 --steps 200 --pipeline 1 --stress --no-cpu-baseline --no-target-leg
run n20_12state --workload mpc_12state_B4096_N20 --steps 200
run n20_12state_serial --workload mpc_12state_B4096_N20 --steps 200 --pipeline 1 --no-cpu-baseline --no-target-leg
run n20_12state_serial_overlap --workload mpc_12state_B4096_N20 --steps 200 --pipeline 1 --overlap --no-cpu-baseline --no-target-leg
run n20_12state_smooth --workload mpc_12state_B4096_N20 --steps 200 --pipeline 1 --smoothing --no-cpu-baseline --no-target-leg
run rollout_12state --workload rollout_12state_B8192_N20 --steps 500 --no-cpu-baseline --no-target-leg
run rollout_6dof --workload rollout_6dof_B8192_N20 --steps 500 --no-cpu-baseline --no-target-leg
run rollout_6dof_budget250 --workload rollout_6dof_B8192_N20 --steps 500 --max-iter 250 --no-cpu-baseline --no-target-leg
run n20_6dof_smooth --workload mpc_6dof_B4096_N20 --steps 200 --pipeline 1 --smoothing --no-cpu-baseline --no-target-leg
run n20_6dof_eps1e5 --workload mpc_6dof_B4096_N20 --steps 100 --pipeline 1 --eps 1e-5 --no-cpu-baseline --no-target-leg
run n50_eps1e5 --workload mpc_6dof_B32768_N50 --steps 5 --eps 1e-5 --no-cpu-baseline --no-target-leg
run n50_one_operator --steps 10 --no-cpu-baseline --no-schedule
run n20_12state_stress --workload mpc_12state_B4096_N20 --steps 100 --pipeline 1 --stress --no-cpu-baseline --no-target-leg
run n20_12state_eps1e5 --workload mpc_12state_B4096_N20 --steps 10 --pipeline 1 --eps 1e-5 --no-cpu-baseline --no-target-leg
run n20_12state_eps1e6 --workload mpc_12state_B4096_N20 --steps 10 --pipeline 1 --eps 1e-6 --no-cpu-baseline --no-target-leg
run n20_12state_eps1e6_inflight --workload mpc_12state_B4096_N20 --steps 30 --eps 1e-6 --no-cpu-baseline --no-target-leg
