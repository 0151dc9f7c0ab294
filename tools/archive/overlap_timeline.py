"""Shows how kernels from different frames overlap in a rocprofv3 --kernel-trace CSV (a 2.5 ms window)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("k_raygen", "k_extend", "k_shade", "k_shadow", "k_accumulate"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = rows[len(rows) * 2 // 3]
t0 = int(mid["Start_Timestamp"])
def short(n):
    for k in ("k_raygen", "k_extend", "k_shade", "k_shadow", "k_accumulate"):
        if k in n: return k[2:6]
busy = []
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if -50 <= s <= 2500:
        print(f"q{r.get('Queue_Id','?'):>3s} {short(r['Kernel_Name']):5s} {s:8.1f} -> {e:8.1f}  ({e - s:7.1f} us)")
