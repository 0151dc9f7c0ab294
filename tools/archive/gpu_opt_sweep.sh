#!/bin/bash
# one renderer option swept at the default pass shape, 240 and 20 steps, one box: tools/gpu_opt_sweep.sh persist_chunk 128 256 384 512
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/sweep; mkdir -p $O; cd $R
K=$1; shift
b() { timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %8.1f Mrays/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" || tail -2 $O/last.err; }
for rep in 1 2; do for v in "$@"; do echo "$K=$v"; b --steps 240 --warmup 24 --opt $K=$v; b --steps 20 --warmup 5 --opt $K=$v; done; done
