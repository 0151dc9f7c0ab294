set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SKIP_TESTS=1 bash tools/gpu_variants_ab.sh pr2048 pr8192 > gpurun_out/packrange_ab.log 2>&1
echo "== serial pass" >> gpurun_out/packrange_ab.log
for v in "" pr2048 pr8192 "" pr2048 pr8192; do MRT_LIB_PATH=${v:+$PWD/metal-raytracing_amd/variants/libmrt_hip_$v.so} python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-latency --no-strict --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   [$v] value', d['value'], 'ms/step', d['ms_per_step'], d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" >> gpurun_out/packrange_ab.log; done
cat gpurun_out/packrange_ab.log
