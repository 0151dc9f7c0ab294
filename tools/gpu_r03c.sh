#!/bin/bash
# round-3 third GPU call: full GPU suite with the Halton table on, A/B of the table, VALU instruction counts per frame
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2; do
  for t in 0 1; do echo "halton_table=$t long"; STEPS=240 WARM=24 b --opt halton_table=$t; echo "halton_table=$t 20"; b --opt halton_table=$t; done
done
echo "table 1x4 serial"; b --opt frames_in_flight=1
echo "no table 1x4 serial"; b --opt frames_in_flight=1 --opt halton_table=0
BENCH_ARGS="" bash tools/pmc_pass.sh r03c_tab "SQ_INSTS_VALU" "VALUBusy VALUUtilization"
BENCH_ARGS="--opt halton_table=0" bash tools/pmc_pass.sh r03c_notab "SQ_INSTS_VALU"
