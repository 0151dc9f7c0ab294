#!/bin/bash
# round 6, call 1: the pruned tree (shade.h split, variants removed) — the GPU suite, then head against the round-5 tree (.r5tree) on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_prune; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
b() { ( cd $1 && timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $EXTRA 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err ); }
for rep in 1 2 3; do
  echo "head 240"; b $R; echo "r5 240"; b $R/.r5tree
  echo "head 20"; STEPS=20 WARM=5 b $R; echo "r5 20"; STEPS=20 WARM=5 b $R/.r5tree
done 2>&1 | tee $O/ab.txt
for rep in 1 2; do
  echo "head dragon4 two-level 240"; EXTRA="--scene dragon4 --sopt instancing=1" b $R; echo "r5 dragon4 two-level 240"; EXTRA="--scene dragon4 --sopt instancing=1" b $R/.r5tree
done 2>&1 | tee -a $O/ab.txt
