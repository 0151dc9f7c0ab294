#!/bin/bash
# builder = 2 (host binned SAH): parity, records per ray, frame rate
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "brute_force or cornell_256 or builder_invariance" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python3 tools/visit_probe.py '{"builder": 1}' '{"builder": 2}' '{"builder": 2, "presplit": 0}' '{"builder": 2, "wide_cost_tri": 1.0}' > $O/visit.log 2>&1; grep -v amdgpu $O/visit.log
b() { python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'], 'build_ms', c['bvh_build_ms'], 'nodes', c['bvh_nodes'], 'sah', c['sah_cost'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "builder 1 long"; b; echo "builder 1 20"; STEPS=20 WARM=5 b
  echo "builder 2 long"; b --builder 2; echo "builder 2 20"; STEPS=20 WARM=5 b --builder 2
done
echo "hostile builder 1"; b --scene dragon_hostile; echo "hostile builder 2"; b --scene dragon_hostile --builder 2
echo "irregular builder 1"; b --scene dragon_irregular; echo "irregular builder 2"; b --scene dragon_irregular --builder 2
