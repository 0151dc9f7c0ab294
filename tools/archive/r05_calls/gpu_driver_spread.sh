cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for k in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver command:', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step; one frame alone', d['latency']['ms_per_frame'], '3-in-flight', d['latency']['reference_like_3_in_flight_ms_per_frame'], 'frac', d['roofline']['frac'], 'strict', d['strict_primary_plus_shadow']['value'])"; done | tee -a gpurun_out/driver_spread.log
