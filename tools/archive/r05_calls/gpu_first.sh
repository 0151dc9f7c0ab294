#!/bin/bash
# round 5, first call: the GPU suite on this round's box, the default bench line, and a sweep of the 8-wide collapse's triangle-cost constant
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_first; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'nodes', d['config']['bvh_nodes'])" || tail -3 $O/last.err; }
for rep in 1 2; do for c in 0.3 0.2 0.45 0.6; do echo "[wide_cost_tri=$c] long"; b "--sopt wide_cost_tri=$c"; echo "[wide_cost_tri=$c] 20"; STEPS=20 WARM=5 b "--sopt wide_cost_tri=$c"; done; done
timeout -k 10 200 python3 tools/stream_lane_use.py 1024 > $O/lane_use.txt 2>&1; cat $O/lane_use.txt
