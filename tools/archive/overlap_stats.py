"""Per-kernel launch durations and concurrency from a rocprofv3 --kernel-trace CSV of a default (overlapped) bench run:
average duration per kernel, and how much of the wall time each kernel had at least one launch resident."""
import csv, sys, glob, re, collections
path = sys.argv[1]
f = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = collections.defaultdict(list)
for r in rows:
    m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
    if not m: continue
    ev[m.group(0)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0 = min(s for v in ev.values() for s, _ in v); t1 = max(e for v in ev.values() for _, e in v)
print(f"wall {1e-3 * (t1 - t0):.0f} us")
for k, v in sorted(ev.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    tot = sum(e - s for s, e in v)
    print(f"{k:28s} calls {len(v):5d}  avg {1e-3 * tot / len(v):8.1f} us  min {1e-3 * min(e - s for s, e in v):8.1f}  max {1e-3 * max(e - s for s, e in v):8.1f}  mean concurrency {tot / (t1 - t0):5.2f}")
