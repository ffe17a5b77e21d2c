# sub-rows of the results table: smoothing on, the stress geometry, eps 1e-5
set -e
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/sr_$name.json 2>gpurun_out/sr_$name.err
python - <<PY
import json;d=json.load(open('gpurun_out/sr_$name.json'))
print('$name', 'solves/s %.4g' % d['value'], 'ms %.4g' % d['ms_per_step'], 'iters mean %.1f max %d' % (d['iters_mean'], d['iters_max']), 'solved %.4f' % d['solved_frac'], 'frac %.3f' % d['roofline']['frac'], d['iters_hist'])
PY
}
run n20_smooth --workload mpc_6dof_B4096_N20 --smoothing
run n20_stress --workload mpc_6dof_B4096_N20 --stress --steps 100
run n20_eps5 --workload mpc_6dof_B4096_N20 --eps 1e-5 --steps 200
run s12_smooth --workload mpc_12state_B4096_N20 --steps 50 --smoothing
run s12_stress --workload mpc_12state_B4096_N20 --steps 50 --stress
run s12_eps5 --workload mpc_12state_B4096_N20 --steps 30 --eps 1e-5
run n50_stress --workload mpc_6dof_B32768_N50 --steps 10 --stress
