#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
bash tools/gpu_results_table.sh > $O/table.log 2>&1; cat $O/table.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python3 -c "
import json
for f in ('bench_default','bench_driver'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], 'ms_per_frame', d.get('ms_per_frame'), 'strict', d.get('strict_primary_plus_shadow',{}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores'), 'parity', d.get('parity'))
    print('  roofline', {k: d['roofline'][k] for k in ('kernel','achieved','frac','traffic','avg_launch_ms')}, d['roofline'].get('serialised_four_frame_launch'), d['roofline'].get('serialised_one_frame_launch'))
    print('  latency', d.get('latency'))
"
python3 tools/tile_scaling_probe.py --steps 20 --warmup 5 --batches 4,8,32 > $O/tile20.log 2>&1; grep -v amdgpu $O/tile20.log
python3 tools/tile_scaling_probe.py --steps 240 --warmup 24 --batches 4,32 > $O/tile240.log 2>&1; grep -v amdgpu $O/tile240.log
