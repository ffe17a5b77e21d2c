# rocprofv3 passes for a round's profiles: kernel trace + stats, FETCH_SIZE, WRITE_SIZE and the SQ
# counters, one run each, per workload (run on the GPU box from the repository root).
#   ROUND=r03 [WORKLOADS="..."] [EXTRA="--max-iter 125"] [TAG=_maxiter125] bash tools/prof_round.sh
# Summaries land in gpurun_out/prof_${ROUND}_sum/ (text + JSON + the bench line of the traced run);
# copy the ones to keep into profiles/.
ROUND=${ROUND:-r03}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
for w in ${WORKLOADS:-mpc_6dof_B262144_N50 mpc_6dof_B4096_N20 mpc_12state_B4096_N20}; do
  D=$R/gpurun_out/prof_$ROUND/$w; mkdir -p $D
  steps=200; case $w in mpc_6dof_B262144_N50) steps=10;; esac
  export PROF_ARGS="--steps $steps --warmup 2 --no-cpu-baseline --no-target-leg $EXTRA"
  A="--workload $w $PROF_ARGS"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D/trace -- python3 $R/bench.py $A > $D.trace.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $D/pmc_fetch -- python3 $R/bench.py $A > $D.fetch.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $D/pmc_write -- python3 $R/bench.py $A > $D.write.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc $C -d $D/pmc_sq -- python3 $R/bench.py $A > $D.sq.log 2>&1 || exit 1
  echo done $w
done
# summaries are made on the box (the databases stay there: only text / JSON travels back)
cd $R
for w in ${WORKLOADS:-mpc_6dof_B262144_N50 mpc_6dof_B4096_N20 mpc_12state_B4096_N20}; do
  steps=200; case $w in mpc_6dof_B262144_N50) steps=10;; esac
  export PROF_ARGS="--steps $steps --warmup 2 --no-cpu-baseline --no-target-leg $EXTRA"
  python3 tools/prof_summary.py gpurun_out/prof_$ROUND/$w gpurun_out/prof_${ROUND}_sum/${ROUND}_$w$TAG $w > gpurun_out/prof_$ROUND/$w.summary.log 2>&1
  grep "^{" gpurun_out/prof_$ROUND/$w.trace.log > gpurun_out/prof_${ROUND}_sum/${ROUND}_${w}${TAG}_bench_line.json
  rm -rf gpurun_out/prof_$ROUND/$w
done
