set -o pipefail
O=gpurun_out/r04final; mkdir -p $O
bash tools/collect_profiles.sh r04 > $O/collect_r04.log 2>&1; tail -3 $O/collect_r04.log | cut -c1-200
for k in 1 2 3; do python bench.py --steps 20 --warmup 5 > $O/bench_driver_$k.json 2> $O/bench_driver_$k.err; python -c "
import json; d=json.loads(open('$O/bench_driver_$k.json').read().strip().splitlines()[-1]); print('driver', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline']['serialised_pass_launch']['frac'], d['cpu_baseline']['value'], d['parity'])"; done
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['strict_primary_plus_shadow']['value'], d['ms_per_frame'])"
