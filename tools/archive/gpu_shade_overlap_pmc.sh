#!/bin/bash
# Why do the shade launches stretch under overlap?  Wave-residency counters of every kernel class in three regimes, one box:
#   serial   one stream, 8-frame passes (every launch alone on the chip)
#   driver   the driver's command: 20 steps after 5, three passes in flight
#   steady   240 steps after 24, six passes in flight
# Per kernel class and regime: dispatches, SQ_WAVES (waves launched), SQ_WAVE_CYCLES (cycles waves were RESIDENT, summed), SQ_BUSY_CYCLES, SQ_WAIT_ANY / SQ_WAIT_INST_ANY /
# SQ_ACTIVE_INST_ANY (resident cycles spent waiting on memory / waiting to issue / issuing), and the L2 hit rate.  A launch whose duration grows under overlap while its waves'
# resident cycles do not is waiting for WAVE SLOTS (held by the persistent traversal waves); one whose resident cycles grow in step is waiting for memory or issue slots.
# usage: tools/gpu_shade_overlap_pmc.sh [OUTDIR]  -> OUTDIR/shade_overlap_pmc.txt (+ the kernel-trace durations of the same three commands)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=${1:-$R/gpurun_out/shade_overlap}; case $O in /*) ;; *) O=$R/$O ;; esac; rm -rf $O; mkdir -p $O
COMMON="--no-cpu-baseline --no-strict --no-latency"
declare -A CMD=( [serial]="--steps 16 --warmup 8 $COMMON --opt frames_in_flight=1 --opt frame_batch=8" [driver]="--steps 20 --warmup 5 $COMMON" [steady]="--steps 240 --warmup 24 $COMMON" )
cd /tmp && export TMPDIR=/tmp
for mode in serial driver steady; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -- python3 $R/bench.py ${CMD[$mode]} $EXTRA > $O/bench_$mode.json 2> $O/trace_$mode.err || { echo "trace $mode failed"; tail -3 $O/trace_$mode.err; exit 1; }
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE VALUBusy VALUUtilization"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${mode}_$i -- python3 $R/bench.py ${CMD[$mode]} $EXTRA > /dev/null 2> $O/pmc_${mode}_$i.err || { echo "pmc $mode $i failed"; tail -3 $O/pmc_${mode}_$i.err; exit 1; }
  done
  echo "$mode done"
done
cd $R
python3 - $O <<'PY' | tee $O/shade_overlap_pmc.txt
import csv, glob, collections, re, sys, json
O = sys.argv[1]
def kname(n):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", n)
    if not m: return None
    t = [x.strip() for x in (m.group(2) or "<>")[1:-1].split(",")]
    if m.group(1) == "k_shade" and len(t) >= 4 and t[3] in ("1", "2", "3"): return "k_shade<primary>"
    return m.group(1)
for mode in ("serial", "driver", "steady"):
    print(f"== {mode}")
    try:
        d = json.loads(open(f"{O}/bench_{mode}.json").read().strip().splitlines()[-1]); print(f"   bench under rocprofv3 --kernel-trace: {d['value']:.0f} Mrays/s, {d['ms_per_step']} ms/step")
    except Exception as e: print("   (no bench line)", e)
    dur = {}
    for f in glob.glob(f"{O}/trace_{mode}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Name"])
            if k: dur[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for f in glob.glob(f"{O}/pmc_{mode}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if not k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
    for k in sorted(agg):
        c = {n: v / max(1, len(nd[(k, n)])) for n, v in agg[k].items()}
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        line = f"   {k:28s} calls {dur.get(k, (0, 0))[0]:4d} avg {dur.get(k, (0, 0))[1]:8.1f} us | waves {c.get('SQ_WAVES', 0):9.0f} resident Mcycles {wc / 1e6:9.2f} busy {c.get('SQ_BUSY_CYCLES', 0) / 1e6:8.2f}"
        if wc: line += f" | of resident: wait mem {c.get('SQ_WAIT_ANY', 0) / wc:5.2f} wait issue {c.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f} issuing {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.2f}"
        if c.get("TCC_REQ_sum"): line += f" | L2 hit {c.get('TCC_HIT_sum', 0) / max(1.0, c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0)):5.2f} req M {c['TCC_REQ_sum'] / 1e6:7.2f}"
        if "VALUBusy" in c: line += f" | VALUBusy {c['VALUBusy']:5.1f} util {c.get('VALUUtilization', 0):5.1f}"
        print(line)
PY
