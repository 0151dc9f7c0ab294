#!/bin/bash
# round 6: per-kernel times of a two-level 8-frame pass ALONE on the chip (one stream), fused TLAS pass against the pass as its own launch, flattened beside them
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06_tl_trace; rm -rf $OUT; mkdir -p $OUT
COMMON="--no-cpu-baseline --no-strict --no-latency --scene dragon4 --steps 32 --warmup 8 --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fused -- python3 $R/bench.py --sopt instancing=1 $COMMON > $OUT/fused.json 2> $OUT/fused.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/unfused -- python3 $R/bench.py --sopt instancing=1 --opt tl_fuse=0 $COMMON > $OUT/unfused.json 2> $OUT/unfused.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/flat -- python3 $R/bench.py $COMMON > $OUT/flat.json 2> $OUT/flat.err
cd $R
for d in fused unfused flat; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("mrt::(anonymous namespace)::", "").replace("void ", "")
    print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} pct {r['Percentage']}")
PY
done > $OUT/summary.txt
cat $OUT/summary.txt
