#!/bin/bash
# round 5: counters of the pooled walk against the stream walk, serialised 8-frame launches (tools/pmc_pass.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_pool_pmc; mkdir -p $O; cd $R
C1="VALUBusy VALUUtilization"; C2="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; C3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
( echo "## stream walk (default)"; BENCH_ARGS="" bash tools/pmc_pass.sh r05s "$C1" "$C2" "$C3"
  echo "## pooled walk (--opt pool=1)"; BENCH_ARGS="--opt pool=1" bash tools/pmc_pass.sh r05p "$C1" "$C2" "$C3"
  echo "## stream walk on 4864 waves"; BENCH_ARGS="--opt wave_slots=4864" bash tools/pmc_pass.sh r05s48 "$C1" "$C2"
  echo "## pooled walk on 4864 waves"; BENCH_ARGS="--opt pool=1 --opt wave_slots=4864" bash tools/pmc_pass.sh r05p48 "$C1" "$C2" ) > $O/pool_pmc.txt 2>&1
grep -v amdgpu.ids $O/pool_pmc.txt | cut -c1-1200
