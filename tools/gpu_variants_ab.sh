#!/bin/bash
# build variants against the head library, same box: tests on each variant (a fast subset), then three alternations at 240 and 20 steps
# usage: tools/gpu_variants_ab.sh NAME [NAME ...]      (metal-raytracing_amd/variants/libmrt_hip_NAME.so, tools/build_variant.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/variants_ab; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $EXTRA 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
[ -n "$SKIP_TESTS" ] || for v in "$@"; do
  echo "== tests on $v"; MRT_LIB_PATH=$V/libmrt_hip_$v.so timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py tests/test_hostile.py -m gpu -x -q 2>&1 | tail -2
done
for rep in 1 2 3; do
  echo "head long"; b; echo "head 20"; STEPS=20 WARM=5 b
  for v in "$@"; do echo "$v long"; MRT_LIB_PATH=$V/libmrt_hip_$v.so b; echo "$v 20"; MRT_LIB_PATH=$V/libmrt_hip_$v.so STEPS=20 WARM=5 b; done
done
