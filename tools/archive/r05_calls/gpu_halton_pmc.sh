set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/halton_pmc.log
for ht in 1 0; do
  echo "#### halton_table=$ht" >> gpurun_out/halton_pmc.log
  BENCH_ARGS="--opt halton_table=$ht" bash tools/pmc_pass.sh ht$ht "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY" "VALUBusy VALUUtilization" "GRBM_GUI_ACTIVE" 2>&1 | grep -v "^k_trace\|^k_accum\|^k_shade " >> gpurun_out/halton_pmc.log
done
cat gpurun_out/halton_pmc.log
