#!/bin/bash
# round-3 sixth GPU call: full GPU suite with pre-splitting on, hostile-mesh rates against the split factor
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
b() { python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'], 'build_ms', c['bvh_build_ms'], 'nodes', c['bvh_nodes'], 'rays/frame', round(c['rays_per_frame']['closest']+c['rays_per_frame']['shadow']))" || tail -3 $O/last.err; }
for ps in 0 1.5 2 4 8; do echo "hostile presplit=$ps"; b --scene dragon_hostile --sopt presplit=$ps; done
for ps in 0 4; do echo "dragon presplit=$ps"; b --sopt presplit=$ps; echo "irregular presplit=$ps"; b --scene dragon_irregular --sopt presplit=$ps; done
echo "hostile presplit=4 20 steps"; STEPS=20 WARM=5 b --scene dragon_hostile
echo "hostile greedy collapse"; b --scene dragon_hostile --sopt wide_collapse=0
