#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/flow4; mkdir -p $O; cd $R
V=$R/metal-raytracing_amd/variants
rc=0

for a in "$@"; do
  for w in ${WAVES:-4 5 6}; do
  echo "== waves $w: $a"
  MRT_LIB_PATH=$V/libmrt_hip_fstats$w.so timeout -k 10 200 python3 bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-latency --no-strict --opt flow=1 $a 2> $O/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"
  grep "flow stats\|flow claims" $O/err.log | tail -2
  done
done
