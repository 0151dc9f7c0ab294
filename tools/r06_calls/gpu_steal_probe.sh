#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_steal; mkdir -p $O; cd $R
for v in wavetimes wavetimes_nosteal; do echo "=== $v"; MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_$v.so timeout -k 10 300 python3 tools/r06_calls/drain_probe.py 2>&1 | grep -v amdgpu.ids | grep -A1 "== dragon:"; done | tee $O/probe_steal.txt
