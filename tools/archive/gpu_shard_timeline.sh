#!/bin/bash
# the driver's 20 steps as rank 0 of eight sees them (1/8 of the tiles, 3 passes of 7 + 7 + 6 frames), on a timeline: tools/gpu_shard_timeline.sh [world]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/shard_timeline; rm -rf $O; mkdir -p $O
W=${1:-8}
cat > $O/run.py <<PY
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, "$R")
import metal_raytracing_amd as mrt
r = mrt.Renderer((1920, 1080), mrt.DragonScene((1920, 1080)), seed=1)
if $W > 1: r.set_shard(0, $W)
for rep in range(3):
    r.draw(5); r.wait(); r.reset_stats()
    t0 = time.perf_counter(); r.draw(20); r.wait(); dt = time.perf_counter() - t0
    st = r.stats; print("wall %.3f ms  %.1f Mrays/s" % (dt * 1e3, (st.closest_rays + st.shadow_rays) / dt / 1e6))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $O/run.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R; grep wall $O/run.log
python3 tools/timeline20.py $O/t 3 > $O/timeline.txt; cat $O/timeline.txt
