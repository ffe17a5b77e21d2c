set -e
cd $GRAFT_REPO_ROOT
WORKLOADS=mpc_6dof_B262144_N50 TAG=_q1 bash tools/prof_r02.sh
grep -n "FETCH\|WRITE\|HBM" gpurun_out/prof_r02_sum/r02_mpc_6dof_B262144_N50_q1_rocprof_summary.txt
cat gpurun_out/prof_r02_sum/r02_mpc_6dof_B262144_N50_q1_bench_line.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['roofline']['frac'])"
