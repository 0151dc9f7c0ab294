set -e
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout -k 10 900 python tools/latency_ab.py reps=3 - stream_stride=1 stream_stride=1,stream_even=100 stream_even=100 stream_stride=1,stream_even=300 > gpurun_out/lat_b.log 2>&1
cat gpurun_out/lat_b.log
