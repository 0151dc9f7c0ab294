#!/bin/bash
# round 5, second call: the suite on the new paths (fat shading records, hit_lds, lds_top), the two probes, and same-box A/Bs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_second; mkdir -p $O; cd $R
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
V=$R/metal-raytracing_amd/variants
for m in 1 2; do MRT_LIB_PATH=$V/libmrt_hip_probe$m.so timeout -k 10 200 python3 tools/stream_level_probe.py $m > $O/probe$m.txt 2>&1; cat $O/probe$m.txt | grep -v amdgpu.ids; done
b() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-240} --warmup ${WARM:-24} --no-cpu-baseline --no-latency --no-strict $1 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'trace', d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
for rep in 1 2 3; do
  for v in "" "--sopt fat_shade=0" "--opt hit_lds=1" "--opt lds_top=1" "--opt lds_top=2" "--opt lds_top=3" "--opt lds_top=4" "--opt hit_lds=1 --opt lds_top=2" "--sopt wide_cost_tri=0.6" "--sopt wide_cost_tri=0.8" "--sopt wide_cost_tri=1.0"; do
    echo "[${v:-default}] long"; b "$v"; echo "[${v:-default}] 20"; STEPS=20 WARM=5 b "$v"
  done
done
# the kernels alone: serialised 8-frame passes
for v in "" "--sopt fat_shade=0" "--opt hit_lds=1" "--opt lds_top=2" "--opt hit_lds=1 --opt lds_top=2"; do echo "[${v:-default}] serialised passes"; STEPS=32 WARM=8 b "$v --opt frames_in_flight=1 --opt frame_batch=8"; done
