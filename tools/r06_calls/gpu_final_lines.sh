#!/bin/bash
# bench.py's own two lines on the final tree, with profiles/r06_summary.json (same csrc hash) in place, + the tables that depend on csrc
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/profiles_r06; mkdir -p $O
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_command.json 2> $O/driver.err || { tail -5 $O/driver.err; exit 1; }
timeout -k 10 400 python3 bench.py > $O/r06_bench_default.json 2> $O/default.err || { tail -5 $O/default.err; exit 1; }
python3 -c "
import json
for f in ('driver_command','default'):
    d=json.loads(open('$O/r06_bench_'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']; print(f, d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], 'rocprof', r.get('avg_launch_ms_rocprof_serialised_pass'), 'live', r['avg_launch_ms'], 'lat', d['latency']['ms_per_frame'], d['latency']['reference_like_3_in_flight_ms_per_frame'])"
