#!/bin/bash
# Round-2, call C: persistent chunk-pulling traversal vs the static split.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell or dragonscene_small or backends or frame_batch or ragged or 1080p_crop or c5_instanced" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'launch_ms', d['roofline']['avg_launch_ms'])"; }
for p in 0 1; do
for cfg in "1 1" "1 4" "1 20" "2 10" "3 1" "3 4" "5 4" "12 4"; do set -- $cfg; echo "persistent=$p fif=$1 batch=$2"; b --opt persistent=$p --opt frames_in_flight=$1 --opt frame_batch=$2; done
done
for c in 128 256 1024 2048; do echo "chunk=$c fif=1 batch=4"; b --opt persist_chunk=$c --opt frames_in_flight=1 --opt frame_batch=4;  echo "chunk=$c fif=5 batch=4"; b --opt persist_chunk=$c --opt frames_in_flight=5 --opt frame_batch=4; done
for p in 0 1; do echo "long persistent=$p"; STEPS=480 WARM=48 b --opt persistent=$p; echo "long persistent=$p fif=4"; STEPS=480 WARM=48 b --opt persistent=$p --opt frames_in_flight=4;  done
