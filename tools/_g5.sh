set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/full_tests.log 2>&1 || { tail -30 gpurun_out/full_tests.log; exit 1; }
tail -3 gpurun_out/full_tests.log
bash tools/bench_lines_r02.sh > gpurun_out/bench_lines.log 2>&1 || { tail gpurun_out/bench_lines.log; exit 1; }
cat gpurun_out/bench_lines.log
