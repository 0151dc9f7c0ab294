#!/bin/bash
# the round's evidence on the final tree, one box: rocprofv3 traces + PMC passes (tools/collect_profiles.sh), then bench.py's own lines (driver command, default)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1; rc=$?; tail -5 gpurun_out/r06_collect.log; [ $rc -eq 0 ] || exit $rc
mkdir -p gpurun_out/profiles_r06
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/profiles_r06/r06_bench_driver_command.json 2> gpurun_out/profiles_r06/driver.err || { tail -5 gpurun_out/profiles_r06/driver.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/profiles_r06/r06_bench_driver_command.json').read().strip().splitlines()[-1]); print('driver:', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'lat', d['latency']['ms_per_frame'], d['latency']['reference_like_3_in_flight_ms_per_frame'], 'cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None, 'parity', d.get('parity'))"
timeout -k 10 400 python3 bench.py > gpurun_out/profiles_r06/r06_bench_default.json 2> gpurun_out/profiles_r06/default.err || { tail -5 gpurun_out/profiles_r06/default.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/profiles_r06/r06_bench_default.json').read().strip().splitlines()[-1]); print('default:', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'strict', d['strict_primary_plus_shadow']['value'])"
