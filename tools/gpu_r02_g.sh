#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02g; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_materials.py tests/test_instancing.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "new rc=$?"; tail -25 $O/pytest_new.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-strict --no-latency "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
echo "default 20"; b
echo "default long"; STEPS=240 WARM=24 b
echo "materials long"; STEPS=240 WARM=24 b --opt materials=1
