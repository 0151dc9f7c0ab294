"""Timeline of the TIMED region of a 20-step bench run from a rocprofv3 --kernel-trace CSV: per launch (stream, kernel, start, duration), relative to the first
launch of the last render() call (the last 5 k_trace_primary launches mark its passes)."""
import csv, sys, glob, re
f = max(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=lambda p: len(open(p).read()))
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if re.search(r"k_(trace|shade|accumulate)", r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 5
def first_of_pass(n):        # the launch a pass starts with: the primary trace, or the shade launch that carries it (fuse_primary)
    m = re.search(r"k_shade<([^>]*)>", n)
    return "k_trace_primary" in n or (m is not None and len(m.group(1).split(",")) >= 4 and m.group(1).split(",")[3].strip() in ("1", "2", "3"))      # k_shade<MATERIALS, CHAIN, PLANES, TRACE0, PAIRS>: TRACE0 != 0
prim = [i for i, r in enumerate(rows) if first_of_pass(r["Kernel_Name"])]
start = prim[-npass]
t0 = int(rows[start]["Start_Timestamp"])
sel = rows[start:]
end = max(int(r["End_Timestamp"]) for r in sel)
print(f"timed region: {len(sel)} launches, {1e-3 * (end - t0):.0f} us")
streams = {}
for r in sel:
    q = r.get("Stream_Id") or r.get("Queue_Id")
    streams.setdefault(q, len(streams))
    m = ("primary+" if first_of_pass(r["Kernel_Name"]) and "k_shade" in r["Kernel_Name"] else "") + re.search(r"k_[a-z_]+", r["Kernel_Name"]).group(0).replace("k_trace_mixed_wide_", "mixed_").replace("k_", "")
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"  lane {streams[q]:2d} {m:16s} {1e-3 * s:8.0f} -> {1e-3 * e:8.0f}  ({1e-3 * (e - s):6.0f} us)  grid {r.get('Grid_Size','?'):>8s} wg {r.get('Workgroup_Size','?')}")
