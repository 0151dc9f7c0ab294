#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_rope_refit; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_refit.py tests/test_build_sizes.py tests/test_instancing.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
