set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_product_paths.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/q_tests.log 2>&1 || { tail -30 gpurun_out/q_tests.log; exit 1; }
tail -3 gpurun_out/q_tests.log
for w in mpc_6dof_B262144_N50 mpc_6dof_B32768_N50 mpc_6dof_B4096_N20 mpc_12state_B4096_N20; do
timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/q_bench.log 2>&1 || { tail -20 gpurun_out/q_bench.log; exit 1; }
grep "^{" gpurun_out/q_bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
WORKLOADS=mpc_6dof_B262144_N50 TAG=_gq bash tools/prof_r02.sh
grep -n "FETCH\|WRITE\|HBM" gpurun_out/prof_r02_sum/r02_mpc_6dof_B262144_N50_gq_rocprof_summary.txt
