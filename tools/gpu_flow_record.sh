#!/bin/bash
# The flow pass (renderer option flow = 1) against the wavefront pipeline, same box: tools/gpu_flow_record.sh -> profiles/r03_flow.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/flow_record; mkdir -p $O; cd $R
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-96} --warmup ${WARM:-12} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   %8.1f Mrays/s  %.4f ms per frame' % (d['value'], d['ms_per_step']))" || tail -3 $O/last.err; }
for cfg in "--frames-in-flight 12" "--frames-in-flight 3" "--frames-in-flight 1" "--frames-in-flight 3 --opt frame_batch=1" "--frames-in-flight 1 --opt frame_batch=1"; do
  echo "== $cfg (lanes x frames per pass: default batch 4)"
  echo " pipeline"; b $cfg || exit 1
  echo " flow"; b $cfg --opt flow=1 || exit 1
done
echo "== the driver's 20 steps"; echo " pipeline"; STEPS=20 WARM=5 b; echo " flow"; STEPS=20 WARM=5 b --opt flow=1; echo " flow, one pass of 20 frames in flight twice"; STEPS=20 WARM=5 b --opt flow=1 --opt frame_batch=20 --frames-in-flight 2
