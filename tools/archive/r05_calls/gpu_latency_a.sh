set -e
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout -k 10 400 python tools/latency_probe.py tile_groups=1 tile_groups=1,stream_stride=1 stream_stride=1 tile_groups=1,persistent=1,pool=1 persistent=1,pool=1 tile_groups=1,stream_stride=1,stream_even=100 tile_groups=1,stream_stride=1,stream_even=400 > gpurun_out/lat_a.log 2>&1
cat gpurun_out/lat_a.log
