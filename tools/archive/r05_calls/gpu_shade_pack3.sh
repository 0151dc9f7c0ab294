#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_shade_pack; mkdir -p $O; cd $R
timeout -k 10 800 python3 -m pytest tests/test_instancing.py tests/test_gpu_parity.py tests/test_materials.py -m gpu -x -q > $O/pytest3.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest3.log
[ $rc -eq 0 ] || { grep -E "Error|error|assert|FAILED" $O/pytest3.log | head -20; exit 1; }
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
for rep in 1 2; do
  for v in "" "--opt shade_pack=0"; do echo "[two-level ${v:-packed}] long"; b "--scene dragon4 --sopt instancing=1 $v"; done
  for v in "" "--opt shade_pack=0"; do echo "[flattened dragon4 ${v:-packed}] long"; b "--scene dragon4 $v"; done
  for v in "" "--opt shade_pack=0"; do echo "[garden 4K ${v:-packed}]"; STEPS=96 WARM=24 b "--scene garden --width 3840 --height 2160 $v"; done
  for v in "" "--opt shade_pack=0"; do echo "[cornell 256 ${v:-packed}]"; b "--scene cornell --width 256 --height 256 $v"; done
  for v in "" "--opt shade_pack=0"; do echo "[C3 4 bounces ${v:-packed}]"; STEPS=64 WARM=8 b "--bounces 4 $v"; done
done
