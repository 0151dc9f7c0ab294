#!/bin/bash
# the round's tables on the final tree (profiles/r06_summary.json in place): bench.py's two lines, BASELINE.md's results table, every rank of 8 timed in turn, the refit probe
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/profiles_r06; mkdir -p $O
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_command.json 2> $O/driver.err || { tail -5 $O/driver.err; exit 1; }
timeout -k 10 400 python3 bench.py > $O/r06_bench_default.json 2> $O/default.err || { tail -5 $O/default.err; exit 1; }
python3 -c "
import json
for f in ('driver_command','default'):
    d=json.loads(open('$O/r06_bench_'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']; print(f, d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], 'rocprof', r.get('avg_launch_ms_rocprof_serialised_pass'), 'live', r['avg_launch_ms'])"
bash tools/gpu_results_table.sh 2>&1 | tee $O/r06_results_table.txt
for i in 1 2 3; do timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-strict 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver again:', d['value'], d['ms_per_step'])"; done | tee $O/r06_driver_spread.txt
{ timeout -k 10 500 python3 tools/tile_scaling_probe.py --scene dragon --batches auto --steps 20 --warmup 5 --worlds 1,8 --all-ranks && timeout -k 10 500 python3 tools/tile_scaling_probe.py --scene dragon --batches auto --steps 240 --warmup 24 --worlds 1,8 --all-ranks && timeout -k 10 600 python3 tools/tile_scaling_probe.py --scene garden --width 3840 --height 2160 --batches auto --steps 20 --warmup 5 --worlds 1,8 --all-ranks && timeout -k 10 600 python3 tools/tile_scaling_probe.py --scene dragon4 --batches auto --steps 16 --warmup 5 --worlds 1,8 --all-ranks; } 2>&1 | grep -v amdgpu.ids | tee $O/r06_tile_scaling_all_ranks.txt
timeout -k 10 500 python3 tools/refit_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_refit_probe.txt
timeout -k 10 200 python3 tools/hostile_rate_check.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_hostile_rate.txt
