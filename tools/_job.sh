set -o pipefail
O=gpurun_out/r04n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_instancing.py -q -m gpu -x > $O/pytest_tl.log 2>&1; rc=$?; tail -3 $O/pytest_tl.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || { grep -n "^FAILED\|^E " $O/pytest_tl.log | head -20; exit 1; }
python bench.py --scene dragon4 --sopt instancing=1 --steps 48 --warmup 12 --no-cpu-baseline --no-strict 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-level:', d['value'], 'one frame alone', d['latency']['ms_per_frame'], d['latency']['kernel_ms_serialised'], '3 in flight', d['latency']['reference_like_3_in_flight_ms_per_frame'])" | tee -a $O/latency_two_level.txt
