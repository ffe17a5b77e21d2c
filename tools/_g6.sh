set -e
cd $GRAFT_REPO_ROOT
bash tools/prof_r02.sh
ls gpurun_out/prof_r02_sum
cat gpurun_out/prof_r02_sum/r02_pmc_traffic.json
