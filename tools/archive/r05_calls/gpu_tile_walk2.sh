#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_tile_walk; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
S="--opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1 --bounces 1"
echo "[default] serialised strict"; STEPS=32 WARM=8 b "$S"
echo "[tile_walk 4 levels] serialised strict"; STEPS=32 WARM=8 b "$S --opt tile_walk=1"
for v in tile2 tile3 tile5; do echo "[$v] serialised strict"; MRT_LIB_PATH=$V/libmrt_hip_$v.so STEPS=32 WARM=8 b "$S --opt tile_walk=1"; done
for rep in 1 2; do
  echo "[default] strict long"; b "--bounces 1"
  echo "[tile_walk 4] strict long"; b "--bounces 1 --opt tile_walk=1"
  for v in tile2 tile3 tile5; do echo "[$v] strict long"; MRT_LIB_PATH=$V/libmrt_hip_$v.so b "--bounces 1 --opt tile_walk=1"; done
done
