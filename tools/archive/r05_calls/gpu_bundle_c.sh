set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "frame_bundle or stream_stride" 2>&1 | tail -5
bash tools/gpu_opt_ab.sh "--opt frame_bundle=1" > gpurun_out/bundle_ab3.log 2>&1
cat gpurun_out/bundle_ab3.log
