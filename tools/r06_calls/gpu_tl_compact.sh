#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_tl_compact; mkdir -p $O; cd $R
timeout -k 10 500 python3 -m pytest tests/test_instancing.py tests/test_refit.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $@ 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "dragon4 flattened 240"; b --scene dragon4
  echo "dragon4 two-level compact 240"; b --scene dragon4 --sopt instancing=1
  echo "dragon4 two-level every-lane 240"; b --scene dragon4 --sopt instancing=1 --opt tl_compact=0
  echo "dragon4 two-level compact 48"; STEPS=48 WARM=8 b --scene dragon4 --sopt instancing=1
  echo "dragon4 two-level every-lane 48"; STEPS=48 WARM=8 b --scene dragon4 --sopt instancing=1 --opt tl_compact=0
done 2>&1 | tee $O/ab.txt
for rep in 1 2; do
  echo "garden 4K two-level compact 48"; STEPS=48 WARM=8 b --scene garden --width 3840 --height 2160 --sopt instancing=1
  echo "garden 4K two-level every-lane 48"; STEPS=48 WARM=8 b --scene garden --width 3840 --height 2160 --sopt instancing=1 --opt tl_compact=0
done 2>&1 | tee -a $O/ab.txt
