#!/bin/bash
# round 5, seventh call: where the pooled walk's time goes — lane statistics, the threshold, the kernel alone
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_seventh; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
for v in poolstats poolstats16; do echo "== $v"; MRT_LIB_PATH=$V/libmrt_hip_$v.so timeout -k 10 200 python3 tools/stream_lane_use.py 1024 2>&1 | grep -v amdgpu.ids; done
echo "== stream walk (head)"; timeout -k 10 200 python3 tools/stream_lane_use.py 1024 2>&1 | grep -v amdgpu.ids
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
S="--opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"
echo "[head stream] serialised"; STEPS=32 WARM=8 b "$S"
echo "[head pool at 48] serialised"; STEPS=32 WARM=8 b "$S --opt pool=1"
for v in pool1 pool16 pool32 pool64; do echo "[$v] serialised"; MRT_LIB_PATH=$V/libmrt_hip_$v.so STEPS=32 WARM=8 b "$S --opt pool=1"; done
echo "[head pool at 48, 8192 wave slots asked] serialised"; STEPS=32 WARM=8 b "$S --opt pool=1 --opt wave_slots=4864"
echo "[head stream, 4864 wave slots] serialised"; STEPS=32 WARM=8 b "$S --opt wave_slots=4864"
