#!/bin/bash
# same-box A/B of two builds of the library: tools/gpu_ab.sh  (variants/libmrt_hip_prev.so vs the in-tree build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab; mkdir -p $O
cd $R
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "prev long"; MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_prev.so STEPS=240 WARM=24 b
  echo "new  long"; STEPS=240 WARM=24 b
  echo "prev 20"; MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_prev.so b
  echo "new  20"; b
done
echo "prev 1x4"; MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_prev.so b --opt frames_in_flight=1
echo "new  1x4"; b --opt frames_in_flight=1
