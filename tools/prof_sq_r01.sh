# SQ counter pass (one rocprofv3 run per workload, 8 SQ counters + 1 GRBM): wave / wait / MFMA / LDS cycles
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
for w in mpc_6dof_B4096_N20 mpc_6dof_B32768_N50 mpc_12state_B4096_N20; do
  D=$R/gpurun_out/prof_r01/$w
  steps=200; case $w in mpc_6dof_B32768_N50) steps=20;; esac
  timeout -s KILL 120 rocprofv3 --pmc $C -d $D/pmc_sq -- python3 $R/bench.py --workload $w --steps $steps --no-cpu-baseline > $D.sq.log 2>&1 || exit 1
  echo done $w
done
