# every bench workload once, one JSON line each into gpurun_out/ba_*.json (no CPU baseline)
set -e
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/ba_$name.json 2>gpurun_out/ba_$name.err
python - <<PY
import json;d=json.load(open('gpurun_out/ba_$name.json'))
print('$name', 'solves/s %.4g' % d['value'], 'ms %.4g' % d['ms_per_step'], 'iters mean %.1f max %d' % (d['iters_mean'], d['iters_max']), 'frac %.3f' % d['roofline']['frac'])
PY
}
run n20 --workload mpc_6dof_B4096_N20
run n20_osqp --workload mpc_6dof_B4096_N20 --check-fine 0
run n50_f5 --workload mpc_6dof_B32768_N50 --steps 20
run n50_f0 --workload mpc_6dof_B32768_N50 --steps 20 --check-fine 0
run n50_f0_tile --workload mpc_6dof_B32768_N50 --steps 20 --check-fine 0 --grid-cap -1
run s12 --workload mpc_12state_B4096_N20 --steps 50
run r12 --workload rollout_12state_B8192_N20 --steps 50
run r12_tile --workload rollout_12state_B8192_N20 --steps 50 --grid-cap -1
run r6 --workload rollout_6dof_B8192_N20 --steps 50
run r6_tile --workload rollout_6dof_B8192_N20 --steps 50 --grid-cap -1
