#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02j; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-latency "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'strict', (d.get('strict_primary_plus_shadow') or {}).get('value'), 'bytes', d['config'].get('triangles'))"; }
for rep in 1 2; do echo "driver 20"; b; echo "long 240"; STEPS=240 WARM=24 b; done
echo "serial 1x4"; b --opt frames_in_flight=1 --opt frame_batch=4 --no-strict
echo "serial 1x1"; b --opt frames_in_flight=1 --opt frame_batch=1 --no-strict
echo "irregular long"; STEPS=240 WARM=24 b --scene dragon_irregular --no-strict
echo "dragon4 long"; STEPS=240 WARM=24 b --scene dragon4 --no-strict
echo "garden 4k"; STEPS=96 WARM=12 b --scene garden --width 3840 --height 2160 --no-strict
