#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_shade_pack; mkdir -p $O; cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed_shade" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
S="--opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"
for v in "" "--opt shade_pack=1"; do echo "[${v:-default}] serialised passes"; STEPS=32 WARM=8 b "$v $S"; done
for rep in 1 2 3; do for v in "" "--opt shade_pack=1"; do echo "[${v:-default}] long"; b "$v"; echo "[${v:-default}] 20"; STEPS=20 WARM=5 b "$v"; done; done
