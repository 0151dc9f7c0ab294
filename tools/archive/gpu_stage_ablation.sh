#!/bin/bash
# What does each stage cost under overlap?  Needs the diagnostics build (tools/build_variant.sh diag "-DMRT_DIAGNOSTICS"): only that library reads MRT_ABLATE
# (1 = no primary launches, 2 = no bounce / shadow traversal launches, 3 = neither; the other kernels run on stale but well-formed queues; images are garbage).
# usage: tools/gpu_stage_ablation.sh   -> gpurun_out/stage_ablation/ablation.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/stage_ablation; mkdir -p $O; cd $R
export MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_diag.so
[ -f $MRT_LIB_PATH ] || { echo "build the diagnostics variant first: tools/build_variant.sh diag \"-DMRT_DIAGNOSTICS\""; exit 1; }
b() { python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/step', d['ms_per_step'], 'avg launch ms', d['roofline']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
for rep in 1 2; do
  echo "all stages, long"; b; echo "all stages, 20"; STEPS=20 WARM=5 b
  echo "no primary, long"; MRT_ABLATE=1 b; echo "no primary, 20"; MRT_ABLATE=1 STEPS=20 WARM=5 b
  echo "no trace, long"; MRT_ABLATE=2 b; echo "no trace, 20"; MRT_ABLATE=2 STEPS=20 WARM=5 b
  echo "neither, long"; MRT_ABLATE=3 b; echo "neither, 20"; MRT_ABLATE=3 STEPS=20 WARM=5 b
done 2>&1 | tee $O/ablation.txt
