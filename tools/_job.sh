set -o pipefail
O=gpurun_out/r04shape; mkdir -p $O
{
python tools/tile_scaling_probe.py --worlds 1,8 --batches 3,4,5,7,10,20 --steps 20 --warmup 5
python tools/tile_scaling_probe.py --worlds 8 --batches 7 --steps 20 --warmup 5 --opt frames_in_flight=3
python tools/tile_scaling_probe.py --worlds 8 --batches 7 --steps 20 --warmup 5 --opt stream_even=100
python tools/tile_scaling_probe.py --worlds 8 --batches 7 --steps 20 --warmup 5 --opt stream_even=400
python tools/tile_scaling_probe.py --worlds 8 --batches 7 --steps 20 --warmup 5 --opt persistent=1
python tools/tile_scaling_probe.py --worlds 8 --batches 7 --steps 20 --warmup 5 --opt fuse_primary=0
python tools/tile_scaling_probe.py --worlds 8 --batches 20 --steps 20 --warmup 5 --opt megakernel=1
} 2>&1 | grep -v amdgpu.ids | tee $O/rank8_shapes.txt
