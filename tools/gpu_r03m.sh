#!/bin/bash
# two-level scenes: node visits ahead of triangle tests inside a BLAS
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
timeout -k 10 600 python3 -m pytest tests/test_instancing.py tests/test_cpp_host_mirror.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
MRT_LIB_PATH=$V/libmrt_hip_bounds.so timeout -k 10 600 python3 -m pytest tests/test_instancing.py -m gpu -x -q > $O/pytest_bounds.log 2>&1; echo "pytest bounds rc=$?"; tail -3 $O/pytest_bounds.log
b() { python3 bench.py --steps ${STEPS:-48} --warmup ${WARM:-12} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'], 'build', c['bvh_build_ms'])" || tail -3 $O/last.err; }
for rep in 1 2; do
  echo "dragon4 flat"; b --scene dragon4
  echo "dragon4 two-level (new)"; b --scene dragon4 --sopt instancing=1
  echo "dragon4 two-level (before)"; MRT_LIB_PATH=$V/libmrt_hip_prev2.so b --scene dragon4 --sopt instancing=1
  echo "garden 4K flat"; b --scene garden --width 3840 --height 2160
  echo "garden 4K two-level"; b --scene garden --width 3840 --height 2160 --sopt instancing=1
done
