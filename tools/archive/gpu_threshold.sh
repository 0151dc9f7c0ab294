#!/bin/bash
# where chunk pulling starts to pay: passes of 1 .. 4 frames on 6 lanes, static even split (persistent=0) against chunk pulling (persistent=1), 120 steps
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for fb in 1 2 3 4 6; do for p in 0 1; do echo -n "frame_batch $fb persistent $p: "; python3 bench.py --no-cpu-baseline --no-latency --no-strict --steps 120 --warmup 12 --opt frame_batch=$fb --opt persistent=$p 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done; done
