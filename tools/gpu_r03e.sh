#!/bin/bash
# round-3 fifth GPU call: bounds-build tests; the driver's command shape (20 steps) against grid size / pass size / lanes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_bounds.so timeout -k 10 600 python3 -m pytest tests/test_instancing.py tests/test_fuzz_geometry.py tests/test_materials.py -m gpu -x -q > $O/pytest_bounds.log 2>&1; echo "pytest (bounds build) rc=$?"; tail -3 $O/pytest_bounds.log
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('   value', d['value'], 'ms/step', d['ms_per_step'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "default 20"; b
  for ws in 2048 3584 5120; do echo "20: wave_slots=$ws"; b --opt wave_slots=$ws; done
  for fb in 2 3 5 7 10; do echo "20: frame_batch=$fb"; b --opt frame_batch=$fb; done
  echo "20: frame_batch=5 wave_slots=3584"; b --opt frame_batch=5 --opt wave_slots=3584
  echo "20: frame_batch=2 wave_slots=2048"; b --opt frame_batch=2 --opt wave_slots=2048
  echo "20: persist_chunk=128"; b --opt persist_chunk=128
  echo "20: persist_chunk=512"; b --opt persist_chunk=512
done
