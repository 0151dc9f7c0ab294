set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for fb in 0 1 2; do
  echo "#### frame_bundle=$fb" >> gpurun_out/bundle_pmc.log
  BENCH_ARGS="--opt frame_bundle=$fb" bash tools/pmc_pass.sh fb$fb "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "VALUBusy VALUUtilization" 2>&1 | grep -v "^k_trace\|^k_accum" >> gpurun_out/bundle_pmc.log
done
cat gpurun_out/bundle_pmc.log
