set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/bc_tests.log 2>&1 || { tail -30 gpurun_out/bc_tests.log; exit 1; }
tail -3 gpurun_out/bc_tests.log
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --no-cpu-baseline > gpurun_out/q_bench.log 2>&1 || { tail -20 gpurun_out/q_bench.log; exit 1; }
grep "^{" gpurun_out/q_bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['config'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['p50_batch_ms'])"
