#!/bin/bash
# two-level scenes on the 8-wide stream traversal: tests, then dragon x4 flattened vs two-level (wide) vs two-level (rope)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/two_level; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_instancing.py -x -q -m gpu -s > $O/tests.log 2>&1 || { echo "tests failed"; head -5 $O/tests.log; tail -15 $O/tests.log; exit 1; }
tail -3 $O/tests.log
b() { timeout -k 10 300 python3 bench.py --scene dragon4 --steps 48 --warmup 12 --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'build', d['config']['bvh_build_ms'])"; }
set -e
echo flat; b
echo two-level wide; b --sopt instancing=1
echo two-level rope; b --sopt instancing=1 --opt wide_bounce=0
