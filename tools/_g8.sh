set -e
cd $GRAFT_REPO_ROOT
for cfg in "mpc_6dof_B32768_N50 1" "mpc_6dof_B32768_N50 2" "mpc_6dof_B32768_N50 3" "mpc_6dof_B262144_N50 1" "mpc_6dof_B262144_N50 2"; do set -- $cfg
timeout -k 10 300 python bench.py --no-cpu-baseline --workload $1 --pipeline $2 > gpurun_out/q_bench.log 2>&1 || { tail -20 gpurun_out/q_bench.log; exit 1; }
grep "^{" gpurun_out/q_bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['batches_in_flight'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['p50_batch_ms'])"
done
