#!/bin/bash
# serialised kernel-class times, dragon x4: flattened vs two-level
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/two_level; mkdir -p $O; cd $R
set -e
b() { timeout -k 10 300 python3 bench.py --scene dragon4 --steps 16 --warmup 4 --frames-in-flight 1 --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], d['roofline']['all_kernels_avg_launch_ms'])"; }
echo flat; b
echo flat primary_wide; b --opt primary_wide=1
echo two-level wide; b --sopt instancing=1
