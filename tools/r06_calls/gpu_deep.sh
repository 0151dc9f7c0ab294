#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_deep; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_deep_tree.py tests/test_gpu_parity.py -m gpu -x -q -s -k "deep or large_leaf" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "wide_layout|passed|failed|Error|assert" $O/pytest.log | tail -20
