#!/bin/bash
# Runs one rocprofv3 --pmc pass per argument group (quoted list of counters), each under its own timeout, on a 4K
# single-stream bench (bulk-dominated), and prints the per-kernel means.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  echo "== pass $i: $grp"
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcx_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --width 3840 --height 2160 > /dev/null 2> $R/gpurun_out/pmcx_$i.err
  echo "rc=$?"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcx_$i 2>&1 | grep -E "k_trace|k_shade" | cut -c1-600
done
