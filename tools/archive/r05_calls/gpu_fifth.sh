#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_fifth; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
MRT_LIB_PATH=$V/libmrt_hip_probe1.so timeout -k 10 200 python3 tools/stream_level_probe.py 1 > $O/probe1.txt 2>&1; grep -v amdgpu.ids $O/probe1.txt
echo "== instancing tests on head (thin pairs + hit uv)"; timeout -k 10 400 python3 -m pytest tests/test_instancing.py -m gpu -x -q 2>&1 | tail -2
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "[two-level head: thin pairs + hit uv] long"; b "--scene dragon4 --sopt instancing=1"
  echo "[two-level no hit uv] long"; MRT_LIB_PATH=$V/libmrt_hip_tlnouv.so b "--scene dragon4 --sopt instancing=1"
done
echo "[two-level head, serialised passes]"; STEPS=32 WARM=8 b "--scene dragon4 --sopt instancing=1 --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"
echo "[two-level no hit uv, serialised passes]"; MRT_LIB_PATH=$V/libmrt_hip_tlnouv.so STEPS=32 WARM=8 b "--scene dragon4 --sopt instancing=1 --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"
