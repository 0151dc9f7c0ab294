#!/bin/bash
# counters of the traversal kernels on dragon x 4, flattened and two-level (serialised 4-frame passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for mode in flat tl; do
  [ $mode = tl ] && S="--sopt instancing=1" || S=""
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "VALUBusy VALUUtilization" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/tlpmc_${mode}_$i -- python3 $R/bench.py --scene dragon4 $S --steps 8 --warmup 4 --no-cpu-baseline --no-strict --no-latency --opt frames_in_flight=1 --opt frame_batch=4 > /dev/null 2> $R/gpurun_out/tlpmc_${mode}_$i.err || { echo "pass failed"; tail -3 $R/gpurun_out/tlpmc_${mode}_$i.err; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, collections, re, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
for mode in ("flat", "tl"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for f in glob.glob(f"{R}/gpurun_out/tlpmc_{mode}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z_]+", r["Kernel_Name"])
            if not m or not ("trace" in m.group(0) or "shade" in m.group(0)): continue
            k = m.group(0) + ("<primary>" if re.search(r"k_shade<[^>]*true>", r["Kernel_Name"]) and r["Kernel_Name"].count(",") >= 4 and re.search(r"k_shade<([^>]*)>", r["Kernel_Name"]).group(1).split(",")[-1].strip() == "true" else "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    print("==", mode)
    for k, d in agg.items():
        print("  ", k, {c: round(v / len(nd[(k, c)]) / (1e6 if v > 1e5 else 1), 1) for c, v in d.items()})
PY
