#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03k; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-strict > $O/bench.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
cat $O/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms/step', d['ms_per_step'])"
python3 $R/tools/timeline20.py $O/trace20 5
