#!/bin/bash
# What does SQ_INSTS_VALU count?  tools/valu_rates launches kernels with a known number of wave64 VALU instructions
# (grid x 2048 iterations x 16 instances + a few dozen of prologue / epilogue per wave); compare with the counter.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_cal; rm -rf $O; mkdir -p $O
[ -x $R/tools/valu_rates ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $R/tools/valu_rates $R/tools/valu_rates.hip || { echo "tools/valu_rates did not build"; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -- $R/tools/valu_rates > /dev/null 2> $O/err.txt || { echo failed; tail -5 $O/err.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_fma", "k_mul", "k_max", "k_cvt_ub0", "k_pk_fma", "k_mov"):
    v = acc.get(k) or {}
    if v:
        waves = v["SQ_WAVES"][0]; insts = v["SQ_INSTS_VALU"][0]
        known = waves * 2048 * 16
        print(f"{k}: SQ_WAVES {waves:.0f}, SQ_INSTS_VALU {insts:.0f} = {insts / known:.3f} x the {known:.0f} instructions of the measured loop")
PY
