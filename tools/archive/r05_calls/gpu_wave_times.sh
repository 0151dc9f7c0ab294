set -e
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
export MRT_LIB_PATH=$PWD/metal-raytracing_amd/variants/libmrt_hip_wavetimes.so
mkdir -p gpurun_out
for args in "1 tile_groups=1" "1 tile_groups=1 hit_lds=0 persistent=1" "8 tile_groups=1"; do
  echo "== wave_times $args" >> gpurun_out/wt.log
  timeout -k 10 120 python tools/archive/wave_times.py $args >> gpurun_out/wt.log 2>&1
done
cat gpurun_out/wt.log
