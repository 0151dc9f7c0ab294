#!/bin/bash
# round-3 fourth GPU call: tests of the validator (+ bounds build), sweeps that probe what bounds the overlapped frame
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_instancing.py tests/test_boundary.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_bounds.so timeout -k 10 600 python3 -m pytest tests/test_instancing.py tests/test_fuzz_geometry.py -m gpu -x -q > $O/pytest_bounds.log 2>&1; echo "pytest (bounds build) rc=$?"; tail -3 $O/pytest_bounds.log
b() { python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2; do
  echo "default"; b
  echo "halton_table=2"; b --opt halton_table=2
  for ws in 2048 3072 4096 5120 6144 7168 14336; do echo "wave_slots=$ws"; b --opt wave_slots=$ws; done
done
echo "cornell"; b --scene cornell
echo "cornell 20"; STEPS=20 WARM=5 b --scene cornell
echo "dragon bounces=1"; b --bounces 1
BENCH_ARGS="--no-latency" bash tools/pmc_pass.sh r03d_notab "SQ_INSTS_VALU" "VALUBusy VALUUtilization"
BENCH_ARGS="--no-latency --opt halton_table=1" bash tools/pmc_pass.sh r03d_tab "SQ_INSTS_VALU"
