#!/bin/bash
# a renderer option against its default, same box: the whole -m gpu suite first, then three alternations at 240 and 20 steps, the strict workload, 3 lanes
# usage: tools/gpu_opt_ab2.sh "--opt fuse_primary=0"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/opt_ab2; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "with [$1] long"; b $1; echo "default long"; b
  echo "with [$1] 20"; STEPS=20 WARM=5 b $1; echo "default 20"; STEPS=20 WARM=5 b
done
echo "with [$1] strict"; b --bounces 1 $1; echo "default strict"; b --bounces 1
echo "with [$1] strict 20"; STEPS=20 WARM=5 b --bounces 1 $1; echo "default strict 20"; STEPS=20 WARM=5 b --bounces 1
echo "with [$1] 3 lanes"; b --frames-in-flight 3 $1; echo "default 3 lanes"; b --frames-in-flight 3
echo "with [$1] one frame alone"; b --frames-in-flight 1 --opt frame_batch=1 $1; echo "default one frame alone"; b --frames-in-flight 1 --opt frame_batch=1
echo "with [$1] irregular"; b --scene dragon_irregular $1; echo "default irregular"; b --scene dragon_irregular
