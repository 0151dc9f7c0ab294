#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d; mkdir -p $O
cd $R
tools/valu_rates > $O/valu_rates2.json 2> $O/valu_rates2.err; echo "valu_rates rc=$?"; python3 -c "
import json; d=json.load(open('$O/valu_rates2.json'))
for k,v in d['rates'].items(): print(f'{k:14s} {v[\"Ginst_per_s\"]:8.1f} G/s  {v[\"cycles_per_inst_per_simd\"]:5.2f} cyc  clk {v[\"clock_GHz\"]:.2f}')"
