#!/bin/bash
# The configurations of BASELINE.md's results table, measured on ONE box in one call:  tools/gpu_results_table.sh
# (each line: value Mrays/s, ms per step; a step that fails ends the script)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/table; mkdir -p $O; cd $R
set -e
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name:', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step, build', d['config']['bvh_build_ms'], 'ms, rays/frame', int(d['config']['rays_per_frame']['closest'] + d['config']['rays_per_frame']['shadow']))"; }
b "C1 cornell 256" --scene cornell --width 256 --height 256 --steps 240 --warmup 24
b "C2 driver 20" --steps 20 --warmup 5
b "C2 default 240" --steps 240 --warmup 24
b "C2 3 lanes x 8" --steps 240 --warmup 24 --frames-in-flight 3
b "C2 4 lanes x 8" --steps 240 --warmup 24 --frames-in-flight 4
b "C2 3 lanes x 4" --steps 240 --warmup 24 --frames-in-flight 3 --opt frame_batch=4
b "C2 round-3 default (rope layout resident, primary rays on the rope walk) 240" --steps 240 --warmup 24 --sopt rope=1 --opt primary_wide=0
b "C2 round-3 default 20" --steps 20 --warmup 5 --sopt rope=1 --opt primary_wide=0
b "C2 irregular 20" --scene dragon_irregular --steps 20 --warmup 5
b "C2 irregular 240" --scene dragon_irregular --steps 240 --warmup 24
b "C2 hostile 20" --scene dragon_hostile --steps 20 --warmup 5
b "C2 hostile 240" --scene dragon_hostile --steps 240 --warmup 24
b "C2 hostile 240, no pre-splitting" --scene dragon_hostile --steps 240 --warmup 24 --sopt presplit=0
b "C2 strict (max_bounces 1) 20" --bounces 1 --steps 20 --warmup 5
b "C2 strict (max_bounces 1) 240" --bounces 1 --steps 240 --warmup 24
b "C2 scene without the 8-wide layout (rope kernels) 240" --steps 240 --warmup 24 --sopt wide=0
b "C3 4 bounces 64 frames" --bounces 4 --steps 64 --warmup 8
b "C4 garden 4K" --scene garden --width 3840 --height 2160 --steps 48 --warmup 8
b "C5 dragon4 flat" --scene dragon4 --steps 48 --warmup 12
b "C5 dragon4 two-level (TLAS pass + BLAS pass)" --scene dragon4 --steps 48 --warmup 12 --sopt instancing=1
b "C5 dragon4 two-level, 240 steps" --scene dragon4 --steps 240 --warmup 24 --sopt instancing=1
b "C5 dragon4 two-level in one loop (tl_pairs=0)" --scene dragon4 --steps 48 --warmup 12 --sopt instancing=1 --opt tl_pairs=0
b "C2 materials" --steps 240 --warmup 24 --opt materials=1
