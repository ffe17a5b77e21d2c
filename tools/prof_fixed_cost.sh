# kernel duration of 1-, 5- and 25-iteration solves (rocprofv3 kernel trace): the fixed cost of a launch
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for it in 1 5 25; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fixed/$it -- python3 $R/tools/iter_cost.py 6dof 20 4096 $it $it > $R/gpurun_out/prof_fixed_$it.log 2>&1 || exit 1
done
