# straggler-bound workloads with an iteration budget of 250 per solve
set -e
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/bb_$name.json 2>gpurun_out/bb_$name.err
python - <<PY
import json;d=json.load(open('gpurun_out/bb_$name.json'))
print('$name', 'solves/s %.4g' % d['value'], 'ms %.4g' % d['ms_per_step'], 'iters mean %.1f max %d' % (d['iters_mean'], d['iters_max']), 'solved %.4f' % d['solved_frac'], 'TF %.1f frac %.3f' % (d['roofline']['achieved'], d['roofline']['frac']))
PY
}
run s12_b250 --workload mpc_12state_B4096_N20 --steps 100 --max-iter 250
run r12_b250 --workload rollout_12state_B8192_N20 --max-iter 250
run r6_b250 --workload rollout_6dof_B8192_N20 --max-iter 250
run r12_b100 --workload rollout_12state_B8192_N20 --max-iter 100
