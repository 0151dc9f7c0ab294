#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d; mkdir -p $O
cd $R
tools/valu_rates > $O/valu_rates.json 2> $O/valu_rates.err; echo "valu_rates rc=$?"; python3 -c "
import json; d=json.load(open('$O/valu_rates.json'))
for k,v in d['rates'].items(): print(f'{k:12s} {v[\"Ginst_per_s\"]:8.1f} G/s  {v[\"cycles_per_inst_per_simd\"]:5.2f} cyc  clk {v[\"clock_GHz\"]:.2f}')"
python3 tools/stream_probe.py > $O/stream_probe.log 2>&1; echo "stream_probe rc=$?"; tail -20 $O/stream_probe.log
tools/pmc_pass.sh r02d "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" > $O/pmc.log 2>&1; echo "pmc rc=$?"; cat $O/pmc.log
