#!/bin/bash
# round 5, fourth call: the whole suite on the new defaults (hit_lds, wide_cost_tri 0.5, fat pairs, tile groups by the draw), the staggered tile groups, the two-level A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_fourth; mkdir -p $O; cd $R
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
V=$R/metal-raytracing_amd/variants
echo "== tests on the thin-pairs variant"; MRT_LIB_PATH=$V/libmrt_hip_tlthin.so timeout -k 10 400 python3 -m pytest tests/test_instancing.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 400 python3 tools/latency_groups.py > $O/latency_groups.txt 2>&1; grep -v amdgpu.ids $O/latency_groups.txt
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "[two-level head] long"; b "--scene dragon4 --sopt instancing=1"; echo "[two-level head] 48"; STEPS=48 WARM=8 b "--scene dragon4 --sopt instancing=1"
  echo "[two-level thin pairs] long"; MRT_LIB_PATH=$V/libmrt_hip_tlthin.so b "--scene dragon4 --sopt instancing=1"; echo "[two-level thin pairs] 48"; MRT_LIB_PATH=$V/libmrt_hip_tlthin.so STEPS=48 WARM=8 b "--scene dragon4 --sopt instancing=1"
  echo "[flattened] long"; b "--scene dragon4"
done
for v in "" "--opt tile_groups=1"; do echo "[two-level head, serialised passes $v]"; STEPS=32 WARM=8 b "--scene dragon4 --sopt instancing=1 --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"; done
echo "[two-level thin, serialised passes]"; MRT_LIB_PATH=$V/libmrt_hip_tlthin.so STEPS=32 WARM=8 b "--scene dragon4 --sopt instancing=1 --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"
for rep in 1 2 3; do echo "[default] long"; b ""; echo "[default] 20"; STEPS=20 WARM=5 b ""; done
