#!/bin/bash
# Round-2, call B: calibration, GPU test suite, (batch, lanes) grid at the driver's 20-step run, shade priority.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
python3 tools/calibrate.py > $O/calibrate.json 2> $O/calibrate.err; echo "calibrate rc=$?"; cat $O/calibrate.json
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'launch_ms', d['roofline']['avg_launch_ms'])"; }
for cfg in "12 4" "5 4" "10 2" "12 1" "4 5" "2 10" "1 20" "3 7" "7 3" "6 4" "8 3"; do set -- $cfg; echo "fif=$1 batch=$2"; b --opt frames_in_flight=$1 --opt frame_batch=$2; done
echo "shade_priority fif=12 batch=4"; b --opt shade_priority=1
echo "shade_priority fif=5 batch=4"; b --opt shade_priority=1 --opt frames_in_flight=5
echo "shade_priority fif=10 batch=2"; b --opt shade_priority=1 --opt frames_in_flight=10 --opt frame_batch=2
echo "long default"; STEPS=480 WARM=48 b
echo "long shade_priority"; STEPS=480 WARM=48 b --opt shade_priority=1
echo "long shade_priority fif 6"; STEPS=480 WARM=48 b --opt shade_priority=1 --opt frames_in_flight=6
echo "long fif 6"; STEPS=480 WARM=48 b --opt frames_in_flight=6
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
