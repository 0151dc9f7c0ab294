#!/bin/bash
# small traversal launches (one frame alone, three one-frame passes in flight, a rank of eight, Cornell 256^2) for a few values of one renderer option:
#   tools/gpu_small_launch.sh stream_even 0 50 100 200
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
K=$1; shift
b() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-latency --no-strict "$@" 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %8.1f Mrays/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"; }
echo "== one frame alone"; timeout -k 10 300 python3 tools/latency_probe.py $(for v in "$@"; do echo -n "$K=$v "; done) 2>&1 | grep -v amdgpu.ids | grep "$K\|^{}"
for v in "$@"; do
  echo "== $K=$v"
  echo " 3 lanes x 1 frame, 60 steps"; b --steps 60 --warmup 12 --frames-in-flight 3 --opt frame_batch=1 --opt $K=$v
  echo " 6 lanes x 2 frames, 120 steps"; b --steps 120 --warmup 12 --opt frame_batch=2 --opt $K=$v
  echo " cornell 256"; b --scene cornell --width 256 --height 256 --steps 240 --warmup 24 --opt $K=$v
  echo " rank 0 of 8 / of 4, 20 steps"; timeout -k 10 200 python3 tools/tile_scaling_probe.py --steps 20 --worlds 8,4 --batches 8 --opt $K=$v 2>&1 | grep -v amdgpu.ids
  echo " default 20 steps"; b --steps 20 --warmup 5 --opt $K=$v
done
