#!/bin/bash
# the stream traversal with every node / packet / instance index checked against its array (-DMRT_DEBUG_BOUNDS) under the tests that build, refit and walk trees
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_bounds; mkdir -p $O; cd $R
MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_bounds.so timeout -k 10 900 python3 -m pytest tests/test_instancing.py tests/test_refit.py tests/test_deep_tree.py tests/test_fuzz_geometry.py tests/test_build_sizes.py tests/test_hostile.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "bounds-checked build: pytest rc=$rc"; tail -4 $O/pytest.log
