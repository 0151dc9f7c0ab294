#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_multi; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_group.py tests/test_bench_contract.py tests/test_boundary.py tests/test_abi_and_host.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $O/pytest.log
