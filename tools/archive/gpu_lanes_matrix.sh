#!/bin/bash
# lanes x frames per pass: 12 x 4 (the default until the end of round 3) against 6 x 8 and 4 x 8 on the benchmark configurations, one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/lanes; mkdir -p $O; cd $R
b() { timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('      %8.1f Mrays/s  %.4f ms/step  lane MB %d x %d' % (d['value'], d['ms_per_step'], d['config']['lane_bytes'] >> 20, d['config']['lanes_used']))" || tail -2 $O/last.err; }
for cfg in "--steps 240 --warmup 24" "--steps 20 --warmup 5" "--bounces 1 --steps 240 --warmup 24" "--bounces 1 --steps 20 --warmup 5" "--scene cornell --width 256 --height 256 --steps 240 --warmup 24" "--scene garden --width 3840 --height 2160 --steps 48 --warmup 8" "--scene dragon4 --steps 48 --warmup 12" "--scene dragon4 --steps 48 --warmup 12 --sopt instancing=1" "--scene dragon_hostile --steps 240 --warmup 24" "--bounces 4 --steps 64 --warmup 8" "--opt materials=1 --steps 240 --warmup 24"; do
  echo "== $cfg"
  for lb in "12 4" "6 8" "4 8"; do set -- $lb; echo "   $1 lanes x $2"; b $cfg --frames-in-flight $1 --opt frame_batch=$2; done
done
