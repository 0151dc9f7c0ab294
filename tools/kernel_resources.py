"""Per-kernel register / scratch / LDS / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stdin or a file).
Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt; python tools/kernel_resources.py res.txt"""
import re, subprocess, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
rows = []
for blk in txt.split("Function Name: ")[1:]:
    name = blk.split("\n")[0].split(" [-R")[0]
    g = lambda k: (re.search(rf"{k}: (\d+)", blk) or [None, "?"])[1]
    try: dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception: dem = name
    dem = re.sub(r"\(.*$", "", dem).replace("mrt::(anonymous namespace)::", "").replace("void ", "")
    rows.append((dem[:90], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize \\[bytes/lane\\]"), g("Occupancy \\[waves/SIMD\\]"), g("LDS Size \\[bytes/block\\]")))
print(f"{'kernel':90s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scr':>5s} {'occ':>4s} {'lds':>6s}")
for r in sorted(set(rows)): print(f"{r[0]:90s} {r[1]:>5s} {r[2]:>5s} {r[3]:>5s} {r[4]:>5s} {r[5]:>4s} {r[6]:>6s}")
