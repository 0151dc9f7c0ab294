#!/bin/bash
# same-box comparison of build variants (tools/build_variant.sh): tools/gpu_variants.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/variants; mkdir -p $O
cd $R
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2; do
  echo "base long"; STEPS=240 WARM=24 b; echo "base 20"; b
  for v in "$@"; do echo "$v long"; MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_$v.so STEPS=240 WARM=24 b; echo "$v 20"; MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_$v.so b; done
done
