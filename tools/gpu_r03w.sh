#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03w; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
b() { python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "default long"; b; echo "default 20"; STEPS=20 WARM=5 b
  echo "shade8 long"; MRT_LIB_PATH=$V/libmrt_hip_shade8.so b; echo "shade8 20"; MRT_LIB_PATH=$V/libmrt_hip_shade8.so STEPS=20 WARM=5 b
done
