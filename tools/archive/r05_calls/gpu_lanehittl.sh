set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/lanehittl_ab.log; : > $L
export MRT_LIB_PATH=$PWD/metal-raytracing_amd/variants/libmrt_hip_lanehittl.so
timeout -k 10 900 python -m pytest tests/test_instancing.py tests/test_gpu_parity.py -x -q -m gpu -k "instanc or two_level or dragon4 or c5 or tlas" 2>&1 | tail -3
unset MRT_LIB_PATH
b() { python3 bench.py --scene dragon4 --sopt instancing=1 --steps $1 --warmup $2 --no-cpu-baseline --no-latency --no-strict $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms/step', d['ms_per_step'], d['roofline']['under_overlap']['all_kernels_avg_launch_ms'])"; }
for rep in 1 2 3; do
  for v in "" lanehittl; do
    export MRT_LIB_PATH=${v:+$PWD/metal-raytracing_amd/variants/libmrt_hip_$v.so}; [ -n "$v" ] || unset MRT_LIB_PATH
    echo "[${v:-head}] 240" >> $L; b 240 24 "" >> $L; echo "[${v:-head}] 48" >> $L; b 48 8 "" >> $L
  done
done
for v in "" lanehittl; do
  export MRT_LIB_PATH=${v:+$PWD/metal-raytracing_amd/variants/libmrt_hip_$v.so}; [ -n "$v" ] || unset MRT_LIB_PATH
  echo "[${v:-head}] serial passes" >> $L; b 32 8 "--opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1" >> $L
done
cat $L
