set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_deep_tree.py -q -m gpu -k "empty or deep" > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || { grep -n "^FAILED\|^E " $O/pytest_gpu.log | head -20; exit 1; }
bash tools/gpu_opt_ab.sh "--sopt rope=1 --opt primary_wide=0" "--sopt rope=1" 2>&1 | tee $O/one_bvh_ab.txt
python tools/build_probe.py --reps 5 2>&1 | grep -v amdgpu.ids | tee $O/build_probe.txt
for s in "" "--sopt presplit=2" "--sopt presplit=1"; do echo "hostile $s"; python bench.py --scene dragon_hostile $s --no-cpu-baseline --no-latency --no-strict 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'wide_layout', c['wide_layout'], 'wide_depth', c['wide_depth'], 'nodes', c['bvh_nodes'], 'scene_bytes', c['scene_bytes'])"; done 2>&1 | tee $O/hostile.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 600 $O/bench_driver.json
