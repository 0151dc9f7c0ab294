#!/bin/bash
# One rocprofv3 --pmc pass per argument (a quoted counter list) over a SERIALISED bench run (one stream, so every dispatch
# runs alone), 8-frame passes (the default pass size), every dispatch full-size (steps and warm-up are multiples of the batch).
# (--no-latency: the latency leg runs three lanes side by side, and the counters of overlapping dispatches include each other's traffic — DESIGN.md §6.82)
# usage: tools/pmc_pass.sh TAG "CTR CTR ..." ["CTR ..."] ...   extra bench args via $BENCH_ARGS
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  echo "== pass $i: $grp"
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc_$i -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-strict --no-latency --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1 $BENCH_ARGS > /dev/null 2> $R/gpurun_out/${TAG}_pmc_$i.err || { echo "pass $i failed"; tail -5 $R/gpurun_out/${TAG}_pmc_$i.err; exit 1; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_$i 2>&1 | cut -c1-900
done
