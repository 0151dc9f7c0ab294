#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_deep; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/r06_calls/deep_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/rate.txt
