#!/bin/bash
# shadow planes (default) against the contribution queue + read-modify-write (shadow_planes=0): tests, then three alternations, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/planes; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "queue long"; b --opt shadow_planes=0; echo "planes long"; b
  echo "queue 20"; STEPS=20 WARM=5 b --opt shadow_planes=0; echo "planes 20"; STEPS=20 WARM=5 b
done
echo "queue strict"; b --bounces 1 --opt shadow_planes=0; echo "planes strict"; b --bounces 1
echo "queue dragon4 two-level"; b --scene dragon4 --sopt instancing=1 --opt shadow_planes=0; echo "planes dragon4 two-level"; b --scene dragon4 --sopt instancing=1
echo "queue 3 lanes"; b --frames-in-flight 3 --opt shadow_planes=0; echo "planes 3 lanes"; b --frames-in-flight 3
