#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02h; mkdir -p $O
cd $R
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-strict --no-latency "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2; do
echo "default"; b
for st in 1 3; do echo "stagger=$st"; b --opt stagger=$st; done
echo "batch 5 fif 4"; b --opt frame_batch=5 --opt frames_in_flight=4
echo "batch 3 stagger 1"; b --opt frame_batch=3 --opt stagger=1
echo "batch 6 stagger 1"; b --opt frame_batch=6 --opt stagger=1
echo "batch 8 stagger 3"; b --opt frame_batch=8 --opt stagger=3
done
