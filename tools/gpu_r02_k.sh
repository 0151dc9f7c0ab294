#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02k; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_instancing.py tests/test_cpp_host_mirror.py -m gpu -x -q > $O/pytest_inst.log 2>&1; echo "instancing rc=$?"; tail -8 $O/pytest_inst.log
b() { python3 bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
echo "dragon4 two-level"; STEPS=64 WARM=8 b --scene dragon4 --sopt instancing=1
echo "dragon two-level"; STEPS=64 WARM=8 b --scene dragon --sopt instancing=1
echo "dragon rope only"; STEPS=64 WARM=8 b --scene dragon --sopt wide=0
python3 - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, '.')
import metal_raytracing_amd as mrt
sc = mrt.InstancedDragonScene((320, 180))
r = mrt.Renderer((320, 180), sc, scene_options={"instancing": 1})
xf = mrt.make_transform([1.0, 0.38, 0.5], [0.0, 2.0, 0.0], 1.2)
ts = []
for k in range(5):
    t0 = time.perf_counter(); r.device_scene.set_instance_transform(7, xf); r.device_scene.commit(); ts.append((time.perf_counter() - t0) * 1e3)
print("TLAS-only commit ms:", [round(t, 3) for t in ts], "full two-level build ms (device):", round(r.device_scene.stats.build_ms, 2))
PY
