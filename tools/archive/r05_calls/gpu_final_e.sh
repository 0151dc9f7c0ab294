#!/bin/bash
# round 5, final evidence on the final tree (packed shade): the suite; C3 / materials A/B of the packing; the rocprofv3 evidence; bench.py's own lines with it in profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_final; mkdir -p $O; cd $R
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2; do for v in "" "--opt shade_pack=0"; do echo "[C3 4 bounces ${v:-packed}]"; STEPS=64 WARM=8 b "--bounces 4 $v"; echo "[materials ${v:-packed}]"; b "--opt materials=1 $v"; done; done
bash tools/archive/r05_calls/gpu_final_c.sh
