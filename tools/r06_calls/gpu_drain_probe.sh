#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_drain; mkdir -p $O; cd $R
MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_wavetimes.so timeout -k 10 300 python3 tools/r06_calls/drain_probe.py 2>&1 | tee $O/probe.txt
