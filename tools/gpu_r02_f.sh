#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02f; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_instancing.py -m gpu -x -q > $O/pytest_inst.log 2>&1; echo "instancing rc=$?"; tail -25 $O/pytest_inst.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-strict --no-latency "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'build', d['config']['bvh_build_ms'])"; }
echo "dragon4 flattened"; STEPS=96 WARM=12 b --scene dragon4
echo "dragon4 two-level"; STEPS=96 WARM=12 b --scene dragon4 --sopt instancing=1
echo "dragon two-level"; STEPS=96 WARM=12 b --scene dragon --sopt instancing=1
echo "dragon rope only (wide=0 scene)"; STEPS=96 WARM=12 b --scene dragon --sopt wide=0
