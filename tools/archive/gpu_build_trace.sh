#!/bin/bash
# kernel trace of ONE scene build (dragon, 885 K triangles): tools/gpu_build_trace.sh -> gpurun_out/build_trace/{stats.csv,timeline.txt}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/build_trace; rm -rf $O; mkdir -p $O
cat > $O/one_build.py <<PY
import os, sys
sys.path.insert(0, "$R")
import metal_raytracing_amd as mrt
ctx = mrt.Context(0)
for _ in range(3):
    d = mrt.DeviceScene(ctx, mrt.SCENES["dragon"]((1920, 1080)), {}); print(d.stats.build_ms); d.close()
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/t -- python3 $O/one_build.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, re
k = sorted(glob.glob("$O/t/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(k)), key=lambda r: int(r["Start_Timestamp"]))
# the last build = after the last k_morton
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_morton") or "k_morton" in r["Kernel_Name"])
# walk back to the first kernel of that build (k_extent_sum / k_split_count come before k_morton)
start = idx
while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 300000: start -= 1
rows = rows[start:]
t0 = int(rows[0]["Start_Timestamp"])
agg = {}
with open("$O/timeline.txt", "w") as f:
    prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        m = re.search(r"(k_\w+|__amd_\w+)", r["Kernel_Name"]); name = m.group(1) if m else r["Kernel_Name"][:40]
        f.write(f"{(s - t0) / 1e3:10.1f} us  +gap {(s - prev_end) / 1e3:8.1f}  dur {(e - s) / 1e3:8.1f}  {name}\n")
        a = agg.setdefault(name, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(0, s - prev_end) / 1e3
        prev_end = max(prev_end, e)
    total = (prev_end - t0) / 1e3
    f.write(f"total {total:.1f} us\n")
print(f"total {total:.1f} us")
for name, (n, d, g) in sorted(agg.items(), key=lambda x: -(x[1][1] + x[1][2])):
    print(f"{name:40s} calls {n:4d}  kernel {d:9.1f} us  gap before {g:9.1f} us")
PY
