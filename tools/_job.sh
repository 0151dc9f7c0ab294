set -o pipefail
O=gpurun_out/r04w6; mkdir -p $O
bash tools/gpu_variants_ab.sh wide6 2>&1 | tee $O/wide6_ab.txt
MRT_LIB_PATH=$PWD/metal-raytracing_amd/variants/libmrt_hip_wide6.so python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-strict 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('wide6: nodes', c['bvh_nodes'], 'depth', c['wide_depth'], 'scene_bytes', c['scene_bytes'], 'serial pass', d['latency']['kernel_ms_serialised_pass'], 'one frame', d['latency']['kernel_ms_serialised'])" | tee -a $O/wide6_ab.txt
python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-strict 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('head : nodes', c['bvh_nodes'], 'depth', c['wide_depth'], 'scene_bytes', c['scene_bytes'], 'serial pass', d['latency']['kernel_ms_serialised_pass'], 'one frame', d['latency']['kernel_ms_serialised'])" | tee -a $O/wide6_ab.txt
