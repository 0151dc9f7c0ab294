#!/bin/bash
# round-3 seventh GPU call: node visits ahead of the triangle tests (MRT_WIDE_SPEC) — suite, A/B, lane accounting; sliver-only pre-splitting
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
b() { python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'], 'build_ms', c['bvh_build_ms'], 'nodes', c['bvh_nodes'])" || tail -3 $O/last.err; }
V=$R/metal-raytracing_amd/variants
for rep in 1 2; do
  echo "nospec long"; MRT_LIB_PATH=$V/libmrt_hip_nospec.so b --sopt presplit=0; echo "nospec 20"; MRT_LIB_PATH=$V/libmrt_hip_nospec.so STEPS=20 WARM=5 b --sopt presplit=0
  echo "spec7 long"; MRT_LIB_PATH=$V/libmrt_hip_spec7.so b --sopt presplit=0; echo "spec7 20"; MRT_LIB_PATH=$V/libmrt_hip_spec7.so STEPS=20 WARM=5 b --sopt presplit=0
  echo "spec6 long"; b --sopt presplit=0; echo "spec6 20"; STEPS=20 WARM=5 b --sopt presplit=0
done
echo "spec6 1x4 serial"; STEPS=20 WARM=4 b --sopt presplit=0 --opt frames_in_flight=1
echo "nospec 1x4 serial"; MRT_LIB_PATH=$V/libmrt_hip_nospec.so STEPS=20 WARM=4 b --sopt presplit=0 --opt frames_in_flight=1
for ps in 0 2 4 8; do echo "hostile presplit=$ps"; b --scene dragon_hostile --sopt presplit=$ps; done
for ps in 0 4; do echo "dragon presplit=$ps"; b --sopt presplit=$ps; done
timeout -k 10 200 python3 tools/stream_probe.py '{"presplit": 0}' > $O/stream_spec.log 2>&1; cat $O/stream_spec.log
MRT_LIB_PATH=$V/libmrt_hip_nospec.so timeout -k 10 200 python3 tools/stream_probe.py '{"presplit": 0}' > $O/stream_nospec.log 2>&1; cat $O/stream_nospec.log
BENCH_ARGS="--no-latency --sopt presplit=0" bash tools/pmc_pass.sh r03g_spec "SQ_INSTS_VALU" "VALUBusy VALUUtilization"
