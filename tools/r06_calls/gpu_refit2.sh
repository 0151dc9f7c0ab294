#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/profiles_r06; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_refit.py tests/test_abi_and_host.py tests/test_build_sizes.py -m gpu -x -q > $O/pytest_refit.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest_refit.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python3 tools/refit_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_refit_probe.txt
