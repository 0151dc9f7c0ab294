#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_halton_lanes; mkdir -p $O; cd $R; V=$R/metal-raytracing_amd/variants
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "halton or frame_bundle or backends" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $@ 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'lane GB', round(d['config']['lane_bytes']*d['config']['lanes_used']/1e9,2))" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  echo "interleaved table 240"; b; echo "planes table 240"; MRT_LIB_PATH=$V/libmrt_hip_hplanes.so b
  echo "interleaved table 20"; STEPS=20 WARM=5 b; echo "planes table 20"; STEPS=20 WARM=5 MRT_LIB_PATH=$V/libmrt_hip_hplanes.so b
done 2>&1 | tee $O/halton_ab.txt
for rep in 1 2; do
  for f in 3 4 5 6; do echo "frames_in_flight $f 240"; b --frames-in-flight $f; echo "frames_in_flight $f 20"; STEPS=20 WARM=5 b --frames-in-flight $f; done
done 2>&1 | tee $O/lanes.txt
BENCH_ARGS="" bash tools/pmc_pass.sh r06_halton_lanes/interleaved "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" 2>&1 | grep -E "k_shade_primary" | tee -a $O/halton_ab.txt
