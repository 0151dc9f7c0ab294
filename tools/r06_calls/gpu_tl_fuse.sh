#!/bin/bash
# round 6: the tree-less TLAS pass of two-level scenes inside the shade kernels (tl_fuse) — parity, then A/B on dragon x 4 as instances (240 and 48 steps), flattened beside it
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_tl_fuse; mkdir -p $O; cd $R; V=$R/metal-raytracing_amd/variants
timeout -k 10 500 python3 -m pytest tests/test_instancing.py tests/test_refit.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "frame_bundle or c5 or dragon4" > $O/pytest2.log 2>&1; rc=$?; echo "pytest2 rc=$rc"; tail -3 $O/pytest2.log
[ $rc -eq 0 ] || exit $rc
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict --scene dragon4 $@ 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "flattened 240"; b
  echo "two-level fused 240"; b --sopt instancing=1
  for v in $VARIANTS; do echo "two-level fused $v 240"; MRT_LIB_PATH=$V/libmrt_hip_$v.so b --sopt instancing=1; done
  echo "two-level unfused 240"; b --sopt instancing=1 --opt tl_fuse=0
  echo "two-level fused 48"; STEPS=48 WARM=8 b --sopt instancing=1
  echo "two-level unfused 48"; STEPS=48 WARM=8 b --sopt instancing=1 --opt tl_fuse=0
done 2>&1 | tee $O/ab.txt
