set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/shard_stride.log; : > $L
for rep in 1 2; do
for o in "" "--opt stream_stride=1 --opt stream_even=100" "--opt stream_stride=1" "--opt stream_stride=1 --opt stream_even=150"; do
  echo "## [$o]" >> $L
  timeout -k 10 300 python tools/tile_scaling_probe.py --worlds 1,2,4,8 --batches auto --steps 20 --all-ranks $o 2>&1 | grep -v amdgpu.ids >> $L
done; done
cat $L
