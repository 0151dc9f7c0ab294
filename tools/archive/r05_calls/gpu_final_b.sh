#!/bin/bash
# round 5, final evidence B: BASELINE.md's table on one box, every rank of N for C2 / C4 / C5
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_final; mkdir -p $O; cd $R
bash tools/gpu_results_table.sh > $O/results_table.txt 2>&1; echo "table rc=$?"; cat $O/results_table.txt
( echo "# every rank r of N timed in turn on ONE GPU (no collective): tools/tile_scaling_probe.py --all-ranks; the draw of a real N-GPU run waits for the slowest rank"
  timeout -k 10 300 python3 tools/tile_scaling_probe.py --all-ranks --worlds 1,8 --batches auto --steps 20 --warmup 5
  timeout -k 10 300 python3 tools/tile_scaling_probe.py --all-ranks --worlds 1,8 --batches auto --steps 240 --warmup 24
  timeout -k 10 300 python3 tools/tile_scaling_probe.py --all-ranks --scene garden --width 3840 --height 2160 --worlds 1,8 --batches auto --steps 20 --warmup 5
  timeout -k 10 300 python3 tools/tile_scaling_probe.py --all-ranks --scene dragon4 --worlds 1,8 --batches auto --steps 16 --warmup 4 ) > $O/tile_scaling_all_ranks.txt 2>&1
grep -v amdgpu.ids $O/tile_scaling_all_ranks.txt
