#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-style rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','roofline','latency','calibration','valu_issue','strict_primary_plus_shadow','cpu_baseline','parity'): print(k, json.dumps(d.get(k))[:700])"
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-strict --no-latency "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'launch_ms', d['roofline']['avg_launch_ms'])"; }
for cfg in "1 1" "1 4" "3 1" "2 10" "5 4" "12 4"; do set -- $cfg; echo "fif=$1 batch=$2"; b --opt frames_in_flight=$1 --opt frame_batch=$2; done
echo "long default"; STEPS=480 WARM=48 b
echo "irregular dragon 20"; b --scene dragon_irregular
echo "irregular dragon long"; STEPS=480 WARM=48 b --scene dragon_irregular
echo "dragon4 long"; STEPS=240 WARM=24 b --scene dragon4
echo "garden 4k long"; STEPS=96 WARM=12 b --scene garden --width 3840 --height 2160
