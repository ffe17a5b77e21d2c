# the bench lines kept under profiles/ (with the CPU baseline), after tools/prof_r01.sh
set -e
for w in mpc_6dof_B4096_N20 mpc_6dof_B32768_N50 mpc_12state_B4096_N20 rollout_12state_B8192_N20 rollout_6dof_B8192_N20; do
  steps=500; case $w in mpc_6dof_B32768_N50) steps=50;; mpc_12state*) steps=100;; esac
  timeout -k 10 400 python bench.py --workload $w --steps $steps > gpurun_out/bench_r01_$w.json 2> gpurun_out/bench_r01_$w.err
  echo done $w
done
