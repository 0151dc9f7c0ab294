#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/parity_probe.py '{"presplit": 0}' '{"presplit": 4}' '{"presplit": 0, "r_persistent": 1}' '{"presplit": 4, "r_persistent": 1, "r_persist_chunk": 64}' '{"presplit": 0, "r_wide_stream": 0}' 2>&1 | grep -v amdgpu.ids
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
