#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_pool_v4; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backends and pooled" 2>&1 | tail -2
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'kernels', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
S="--opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"
for rep in 1 2; do
echo "[stream, 4864 waves] serialised"; STEPS=32 WARM=8 b "$S --opt wave_slots=4864"
echo "[pool, 4864 waves] serialised"; STEPS=32 WARM=8 b "$S --opt pool=1 --opt wave_slots=4864"
done
for rep in 1 2 3; do for v in "" "--opt pool=1" "--opt pool=1 --opt wave_slots=3648"; do echo "[${v:-default}] long"; b "$v"; echo "[${v:-default}] 20"; STEPS=20 WARM=5 b "$v"; done; done
C1="VALUBusy VALUUtilization"; C3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"
BENCH_ARGS="--opt pool=1 --opt wave_slots=4864" bash tools/pmc_pass.sh r05p4 "$C1" "$C3" 2>&1 | grep "k_trace\|pass" | cut -c1-400
