#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_ninth; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2; do
  echo "[stream] long"; b ""
  echo "[pool] long"; b "--opt pool=1"
  for v in at16 at48 rf8 rf24 q512; do echo "[pool_$v] long"; MRT_LIB_PATH=$V/libmrt_hip_pool_$v.so b "--opt pool=1"; done
  for ws in 3072 3648 4864; do echo "[pool wave_slots=$ws] long"; b "--opt pool=1 --opt wave_slots=$ws"; done
done
