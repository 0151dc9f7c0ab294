#!/bin/bash
# dragon x 4: flattened against two-level in one loop (tl_pairs = 0) against two-level binned (TLAS pass + BLAS pass, default), same box, three alternations at 48 and 240 steps;
# then rocprofv3 kernel times of one binned and one one-loop run.   usage: tools/gpu_two_level_binned_ab.sh [OUTDIR]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=${1:-$R/gpurun_out/two_level_binned}; case $O in /*) ;; *) O=$R/$O ;; esac; mkdir -p $O; cd $R
b() { timeout -k 10 300 python3 bench.py --scene dragon4 --no-cpu-baseline --no-latency --no-strict "$@" 2> $O/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %8.1f Mrays/s  %.4f ms/step' % (d['value'], d['ms_per_step']), d['roofline']['all_kernels_avg_launch_ms'])" || tail -3 $O/last.err; }
{
for rep in 1 2 3; do
  echo "flattened 48"; b --steps 48 --warmup 12
  echo "two-level one loop 48"; b --steps 48 --warmup 12 --sopt instancing=1 --opt tl_pairs=0
  echo "two-level binned 48"; b --steps 48 --warmup 12 --sopt instancing=1
  echo "two-level one loop 240"; b --steps 240 --warmup 24 --sopt instancing=1 --opt tl_pairs=0
  echo "two-level binned 240"; b --steps 240 --warmup 24 --sopt instancing=1
done
} 2>&1 | tee $O/two_level_binned_ab.txt
cd /tmp && export TMPDIR=/tmp
for mode in binned oneloop; do
  [ $mode = oneloop ] && X="--opt tl_pairs=0" || X=""
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -- python3 $R/bench.py --scene dragon4 --sopt instancing=1 $X --steps 32 --warmup 8 --no-cpu-baseline --no-latency --no-strict --opt frames_in_flight=1 > /dev/null 2> $O/trace_$mode.err || { echo "trace $mode failed"; tail -3 $O/trace_$mode.err; exit 1; }
  echo "== serialised 8-frame passes, $mode: kernel, calls, average us"
  python3 - $O/trace_$mode <<'PY'
import csv, glob, sys, re
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Name"])
        if m and any(x in m.group(1) for x in ("trace", "shade", "tl_", "accumulate")): print(f"   {m.group(1) + (m.group(2) or ''):60s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:10.1f}")
PY
done 2>&1 | tee -a $O/two_level_binned_ab.txt
