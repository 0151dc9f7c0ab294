#!/bin/bash
# same-box A/B of a renderer option: tools/gpu_opt_ab.sh "--opt name=0" "--opt name=1" ...   (default bench vs each argument, three alternations)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in "" "$@"; do
    echo "[${v:-default}] long"; STEPS=240 WARM=24 b "$v"; echo "[${v:-default}] 20"; b "$v"
  done
done
for v in "" "$@"; do echo "[${v:-default}] 1 lane x 4"; STEPS=48 WARM=8 b "$v --frames-in-flight 1"; done
