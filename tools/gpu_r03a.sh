#!/bin/bash
# round-3 first GPU call: the GPU suite on the new build, then same-box A/B of (prev | scaled node test | collapse variants)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'], 'build_ms', c.get('bvh_build_ms'), 'nodes', c.get('bvh_nodes'))" || tail -3 $O/last.err; }
V=$R/metal-raytracing_amd/variants
for rep in 1 2; do
  echo "prev long"; MRT_LIB_PATH=$V/libmrt_hip_prev.so STEPS=240 WARM=24 b; echo "prev 20"; MRT_LIB_PATH=$V/libmrt_hip_prev.so b
  echo "unscaled greedy long"; MRT_LIB_PATH=$V/libmrt_hip_unscaled.so STEPS=240 WARM=24 b --sopt wide_collapse=0
  echo "scaled greedy long"; STEPS=240 WARM=24 b --sopt wide_collapse=0; echo "scaled greedy 20"; b --sopt wide_collapse=0
  for ct in 0.15 0.3 0.5 1.0; do echo "scaled dp tri=$ct long"; STEPS=240 WARM=24 b --sopt wide_cost_tri=$ct; echo "scaled dp tri=$ct 20"; b --sopt wide_cost_tri=$ct; done
done
python3 tools/wide_fill.py dragon > $O/fill.log 2>&1; cat $O/fill.log
