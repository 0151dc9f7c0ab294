R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
b() { python3 bench.py --scene dragon4 --sopt instancing=1 --steps 48 --warmup 12 --no-cpu-baseline --no-latency --no-strict 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2 3; do echo prev; MRT_LIB_PATH=$R/metal-raytracing_amd/variants/libmrt_hip_prev.so b; echo new; b; done
