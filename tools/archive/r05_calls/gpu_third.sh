#!/bin/bash
# round 5, third call: tile groups (tests + the small-launch regimes), wide_cost_tri on the other scenes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_third; mkdir -p $O; cd $R
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_group.py tests/test_instancing.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|error|assert" $O/pytest.log | head -20; exit 1; }
timeout -k 10 400 python3 tools/latency_groups.py > $O/latency_groups.txt 2>&1; grep -v amdgpu.ids $O/latency_groups.txt
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'nodes', d['config']['bvh_nodes'])" || tail -3 $O/last.err; }
for rep in 1 2; do
  for sc in "--scene garden --width 3840 --height 2160" "--scene dragon4" "--scene dragon_hostile" "--scene dragon_irregular" "--scene cornell --width 256 --height 256"; do
    for c in 0.3 0.5; do echo "[$sc wide_cost_tri=$c] long"; STEPS=96 WARM=24 b "$sc --sopt wide_cost_tri=$c"; done
  done
done
