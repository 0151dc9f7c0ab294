set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tl_trace; rm -rf $OUT; mkdir -p $OUT
COMMON="--no-cpu-baseline --no-strict --no-latency"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tl -- python3 $R/bench.py --scene dragon4 --sopt instancing=1 --steps 32 --warmup 8 $COMMON --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1 > $OUT/tl.json 2> $OUT/tl.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/flat -- python3 $R/bench.py --scene dragon4 --steps 32 --warmup 8 $COMMON --opt frames_in_flight=1 --opt frame_batch=8 --opt tile_groups=1 > $OUT/flat.json 2> $OUT/flat.err
cd $R
for d in tl flat; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; cut -d, -f1-8 $f | cut -c1-200 | head -14; done > $OUT/summary.txt
cat $OUT/summary.txt
