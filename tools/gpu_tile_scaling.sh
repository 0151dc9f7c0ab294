#!/bin/bash
# One rank of N on one GPU (no collective), for the BASELINE configs that name 8 GPUs: tools/gpu_tile_scaling.sh OUTDIR
#   C4 garden 3840x2160 and C5 dragon x 4 (1920x1080, 16 frames = spp 16), N = 1/2/4/8, over the driver's 20 frames (C5: 16) and over 240,
#   at the pass size the sharded renderers choose.  Every probe is its own process; a failed one ends the script.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=${1:-$R/gpurun_out/tile_scaling}; mkdir -p $O; cd $R
run() { echo "== $*"; timeout -k 10 400 python3 tools/tile_scaling_probe.py "$@" || exit 1; }
{
run --scene garden --width 3840 --height 2160 --batches auto --steps 20 --warmup 5
run --scene garden --width 3840 --height 2160 --batches auto --steps 240 --warmup 24
run --scene dragon4 --batches auto --steps 16 --warmup 5
run --scene dragon4 --batches auto --steps 240 --warmup 24
run --scene dragon --batches auto --steps 20 --warmup 5
run --scene dragon --batches auto --steps 240 --warmup 24
} 2>&1 | tee $O/tile_scaling.txt
