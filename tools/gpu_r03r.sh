#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
b() { python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "default long"; b; echo "default 20"; STEPS=20 WARM=5 b
  echo "uncached long"; b --opt queue_uncached=1; echo "uncached 20"; STEPS=20 WARM=5 b --opt queue_uncached=1
done
