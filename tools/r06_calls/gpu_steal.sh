#!/bin/bash
# round 6: subtree stealing in the drain phase of the static split's kernel — parity first (a fast subset, then everything), then the small-launch regimes against the same tree without it
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_steal; mkdir -p $O; cd $R; V=$R/metal-raytracing_amd/variants
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backends or ragged or empty" > $O/pytest_fast.log 2>&1; rc=$?; echo "pytest fast rc=$rc"; tail -3 $O/pytest_fast.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  echo "== steal"; timeout -k 10 300 python3 tools/latency_ab.py reps=2 - 2>&1 | grep -v "rep \|amdgpu"
  echo "== no steal"; MRT_LIB_PATH=$V/libmrt_hip_nosteal.so timeout -k 10 300 python3 tools/latency_ab.py reps=2 - 2>&1 | grep -v "rep \|amdgpu"
done 2>&1 | tee $O/latency_ab.txt
