"""Prints the per-launch timeline of the last full frame from a rocprofv3 --kernel-trace CSV (run bench.py with
--frames-in-flight 1 so that a frame's launches do not interleave with its neighbours')."""
import csv, sys, glob, re
path = sys.argv[1]
f = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"k_[a-z0-9_]+", n)
    return m.group(0) if m else n[-30:]
first = ("k_raygen", "k_trace_primary", "k_path")
idx = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith(first)]
start = idx[-2] if len(idx) > 1 else 0
end = idx[-1] if len(idx) > 1 else len(rows)
t0 = int(rows[start]["Start_Timestamp"]); prev_end = t0; tot = {}
for r in rows[start:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    print(f"{n:28s} start={(s - t0) / 1e3:9.1f}us dur={(e - s) / 1e3:8.1f}us gap={(s - prev_end) / 1e3:6.1f}us vgpr={r.get('VGPR_Count', '?')} lds={r.get('LDS_Block_Size', '?')}")
    prev_end = e; tot[n] = tot.get(n, 0) + (e - s)
print("frame total us:", (prev_end - t0) / 1e3, {k: round(v / 1e3, 1) for k, v in tot.items()})
