#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/parity_probe.py '{}' '{"r_persistent": 1}' '{"r_throughput_chain": 0}' '{"r_wide_bounce": 0}' '{"r_fused": 0}' '{"r_max_bounces": 2}' 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
b() { python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'], 'lane MB', c['lane_bytes']>>20)" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "thr queue long"; b --opt throughput_chain=0; echo "thr queue 20"; STEPS=20 WARM=5 b --opt throughput_chain=0
  echo "chain long"; b; echo "chain 20"; STEPS=20 WARM=5 b
done
