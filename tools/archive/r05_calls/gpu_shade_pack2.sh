#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_shade_pack; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_all.log
[ $rc -eq 0 ] || { grep -E "Error|error|assert|FAILED" $O/pytest_all.log | head -20; exit 1; }
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
for rep in 1 2; do
  echo "[range 2048 (head)] long"; b ""; echo "[range 2048 (head)] 20"; STEPS=20 WARM=5 b ""
  for r in 1024 4096 8192; do echo "[range $r] long"; MRT_LIB_PATH=$V/libmrt_hip_pack$r.so b ""; echo "[range $r] 20"; MRT_LIB_PATH=$V/libmrt_hip_pack$r.so STEPS=20 WARM=5 b ""; done
  echo "[unpacked] long"; b "--opt shade_pack=0"
  echo "[pack + pool + tile_walk] long"; b "--opt pool=1 --opt tile_walk=1"; echo "[pack + pool + tile_walk] 20"; STEPS=20 WARM=5 b "--opt pool=1 --opt tile_walk=1"
  echo "[pack + tile_walk] long"; b "--opt tile_walk=1"
done
