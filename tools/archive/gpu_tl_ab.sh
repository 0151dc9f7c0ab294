#!/bin/bash
# two-level variants, same box: dragon x 4 with instancing = 1, three alternations; the instancing tests on each variant first
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/tl_ab; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
b() { timeout -k 10 300 python3 bench.py --scene dragon4 --steps ${STEPS:-96} --warmup 12 --no-cpu-baseline --no-latency --no-strict --sopt instancing=1 "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for v in "$@"; do
  echo "== tests on $v"; MRT_LIB_PATH=$V/libmrt_hip_$v.so timeout -k 10 400 python3 -m pytest tests/test_instancing.py tests/test_hostile.py -m gpu -x -q 2>&1 | tail -2
done
for rep in 1 2 3; do
  echo "head"; b
  for v in "$@"; do echo "$v"; MRT_LIB_PATH=$V/libmrt_hip_$v.so b; done
done
