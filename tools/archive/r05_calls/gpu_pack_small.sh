#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_shade_pack; mkdir -p $O; cd $R
for v in "" "shade_pack=0"; do echo "== latency_groups $v"; timeout -k 10 300 python3 tools/latency_groups.py $v 2>&1 | grep -v amdgpu.ids | head -16; done
