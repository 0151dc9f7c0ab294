"""Summarises rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = next((s for s in ("k_trace_primary", "k_trace_mixed_wide_persist", "k_trace_mixed_wide_stream", "k_trace_mixed_wide", "k_trace_mixed", "k_tl_top", "k_tl_blas", "k_shade_primary", "k_shade_pack", "k_shade", "k_accumulate") if s in n), None)
        if not k: continue
        if k == "k_shade_pack": k = "k_shade"          # the shades of the bounce queues, packed or not: one class (k_shade_primary, bounce 0 with the primary walk inside, is its own)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())}, "dispatches", max(len(v) for v in d.values()))
