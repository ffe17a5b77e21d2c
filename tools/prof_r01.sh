export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for w in mpc_6dof_B4096_N20 mpc_6dof_B32768_N50 mpc_12state_B4096_N20; do
  D=$R/gpurun_out/prof_r01/$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D/trace -- python3 $R/bench.py --workload $w --steps 200 --no-cpu-baseline > $D.trace.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $D/pmc_fetch -- python3 $R/bench.py --workload $w --steps 200 --no-cpu-baseline > $D.fetch.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $D/pmc_write -- python3 $R/bench.py --workload $w --steps 200 --no-cpu-baseline > $D.write.log 2>&1 || exit 1
  echo done $w
done
