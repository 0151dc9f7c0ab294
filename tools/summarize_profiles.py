"""Turns the raw rocprofv3 output of tools/collect_profiles.sh into the small summaries kept under profiles/ (copied there by hand from
gpurun_out/profiles_<tag>/ after the GPU call)."""
import collections, csv, glob, json, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
os.makedirs(P, exist_ok=True)
KERNELS = ("k_trace_primary", "k_trace_mixed_wide_persist", "k_trace_mixed_wide_stream", "k_trace_mixed_wide", "k_trace_mixed", "k_shade", "k_shade_primary", "k_accumulate", "k_accumulate_planes")


def kname(n):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", n)
    if not m:
        return n.split("(")[0]
    targs = [t.strip() for t in (m.group(2) or "<>")[1:-1].split(",")]
    if m.group(1) == "k_shade_pack":
        return "k_shade"                  # the shades of the bounce queues (bounces >= 1), packed or not: one class; k_shade_primary<WALK> (bounce 0 with the primary walk inside) is its own
    if m.group(1).endswith("_x"):
        return m.group(1)[:-2]            # the launches with the hit words in LDS (renderer option hit_lds, default): the same kernel class as the form without them
    return m.group(1)


sys.path.insert(0, ROOT)
import bench
res = {"tag": tag, "csrc_sha256": bench.csrc_hash()}      # bench.py shows these counters only for the source tree they were collected on
for mode in ("driver", "serial", "serial_pass"):
    ks = glob.glob(out + f"/trace_{mode}/**/*kernel_stats.csv", recursive=True)
    if not ks:
        continue
    rows = list(csv.DictReader(open(max(ks, key=os.path.getmtime))))      # gpurun merges into an existing directory: take the newest run
    with open(os.path.join(P, f"{tag}_kernel_stats_{mode}.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            f.write(f"\"{kname(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    res[f"kernel_avg_us_{mode}"] = {kname(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows if kname(r["Name"]) in KERNELS}
    res[f"kernel_calls_{mode}"] = {kname(r["Name"]): int(r["Calls"]) for r in rows if kname(r["Name"]) in KERNELS}
    try:
        res[f"bench_{mode}_under_rocprof"] = json.loads(open(out + f"/bench_{mode}_under_rocprof.json").read().strip().splitlines()[-1])
    except Exception as e:
        res[f"bench_{mode}_under_rocprof"] = str(e)


def pmc_all():
    """{kernel: {counter: mean per dispatch}} over every pmc_* directory; the runs use full passes (bench.PASS_FRAMES frames) with warm-up = one pass, so every dispatch is full size"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    newest = {}
    for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):      # one (the newest) file per pass directory
        d = f[len(out):].split(os.sep)[1]
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]): newest[d] = f
    for f in newest.values():
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k in KERNELS:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}, {k: {c: len(v) for c, v in d.items()} for k, d in acc.items()}


pm, pn = pmc_all()
res["pmc_mean_per_dispatch"] = pm
res["pmc_frames_per_dispatch"] = bench.PASS_FRAMES
res["pmc_dispatches"] = {k: max(v.values()) for k, v in pn.items()}
# HBM traffic per dispatch.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of wide coalesced reads -> x2
# (units: KB = 1024 B); other access widths are uncalibrated, so this is an upper estimate for gathers.
res["hbm_traffic_bytes_per_launch"] = {k: {"fetch_raw_KB": d.get("FETCH_SIZE"), "write_KB": d.get("WRITE_SIZE"),
                                           "bytes_corrected": (2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0)) * 1024 if "FETCH_SIZE" in d else None} for k, d in pm.items()}
res["valu_busy_pct"] = {k: d.get("VALUBusy") for k, d in pm.items()}
res["valu_lane_utilization_pct"] = {k: d.get("VALUUtilization") for k, d in pm.items()}
# VALU wave-instructions per FRAME: per pass of B frames there is 1 primary, max_bounces shade, max_bounces trace and 1 accumulate dispatch
B, BOUNCES = bench.PASS_FRAMES, 3
per_pass = {"k_trace_primary": 1, "k_shade_primary": 1, "k_shade": BOUNCES - 1 if "k_shade_primary" in pm else BOUNCES, "k_trace_mixed_wide_persist": BOUNCES, "k_trace_mixed_wide_stream": BOUNCES, "k_accumulate": 1, "k_accumulate_planes": 1}
insts = sum(pm[k].get("SQ_INSTS_VALU", 0.0) * n for k, n in per_pass.items() if k in pm)
if insts:
    res["valu_wave_insts_per_frame"] = insts / B
    # cycle-weighted estimate with the measured issue costs (tools/valu_rates.hip): fp32 add/mul/fma 2.3 cycles, transcendental 8.2, conversions 4.2,
    # INT32 (and/or/xor/add 2.5, shifts / bit-field / 24-bit multiplies 4.2) taken at 3.3, everything else (min/max, compares, selects ...) 4.2
    cyc = 0.0
    for k, n in per_pass.items():
        if k not in pm:
            continue
        d = pm[k]
        full = d.get("SQ_INSTS_VALU_ADD_F32", 0) + d.get("SQ_INSTS_VALU_MUL_F32", 0) + d.get("SQ_INSTS_VALU_FMA_F32", 0)
        trans, cvt, i32 = d.get("SQ_INSTS_VALU_TRANS_F32", 0), d.get("SQ_INSTS_VALU_CVT", 0), d.get("SQ_INSTS_VALU_INT32", 0)
        rest = max(0.0, d.get("SQ_INSTS_VALU", 0) - full - trans - cvt - i32)
        cyc += n * (2.3 * full + 8.2 * trans + 4.2 * cvt + 3.3 * i32 + 4.2 * rest)
    res["valu_cycles_per_frame"] = cyc / B
    res["valu_cycles_per_frame_note"] = "SIMD cycles of VALU issue per frame, estimated from the SQ_INSTS_VALU_* mix and the measured per-instruction costs; divide by 1024 SIMDs x shader clock for the VALU-bound frame time"
for name in ("valu_rates", "calibrate"):
    try:
        res[name] = json.loads(open(out + f"/{name}.json").read().strip().splitlines()[-1])
    except Exception as e:
        res[name] = str(e)
json.dump(res, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("bench_") and k not in ("valu_rates", "pmc_mean_per_dispatch")}, indent=1)[:6000])
