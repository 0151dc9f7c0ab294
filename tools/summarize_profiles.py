"""Turns the raw rocprofv3 output of tools/collect_profiles.sh into the small summaries kept under profiles/."""
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)      # merged back by gpurun; copy into profiles/ afterwards
os.makedirs(P, exist_ok=True)
res = {}
ks = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    with open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            import re
            mm = re.search(r"(k_[a-z0-9_]+)", r["Name"]); name = mm.group(1) if mm else r["Name"].split("(")[0]
            f.write(f"\"{name}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    for r in rows:
        for k in ("k_trace_primary", "k_trace_mixed_wide_stream", "k_trace_mixed_wide", "k_trace_mixed", "k_extend", "k_shadow", "k_shade", "k_raygen", "k_accumulate"):
            if k + "(" in r["Name"] or r["Name"].endswith(k):
                res.setdefault("kernel_avg_us", {})[k] = float(r["AverageNs"]) / 1e3
                res.setdefault("kernel_calls", {})[k] = int(r["Calls"])
def pmc(dirname, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + f"/{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            for k in ("k_trace_primary", "k_trace_mixed_wide_stream", "k_trace_mixed_wide", "k_trace_mixed", "k_extend", "k_shadow", "k_shade", "k_raygen", "k_accumulate"):
                if k + "(" in r["Kernel_Name"]: acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
res["valu_busy_pct"] = pmc("pmc_valu", "VALUBusy")                 # % of cycles the VALU is issuing (kernels serialised by the profiler)
res["valu_lane_utilization_pct"] = pmc("pmc_valu", "VALUUtilization")   # % of lanes active in an average VALU instruction
# units: KB (1024 B) per dispatch.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes
# of wide coalesced reads -> x2; other access widths are uncalibrated, so this is an upper estimate for gathers.
res["hbm_traffic_bytes_per_launch"] = {k: {"fetch_raw_KB": fetch.get(k), "write_KB": write.get(k),
                                           "bytes_corrected": (2 * fetch.get(k, 0) + write.get(k, 0)) * 1024 if k in fetch else None} for k in set(fetch) | set(write)}
# VALU wave-instructions per frame: every dispatch of the render kernels in the 10-frame counter run (8 steps + 2 warm-up)
insts = 0.0
for f in glob.glob(out + "/pmc_insts/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU" and any(k in r["Kernel_Name"] for k in ("k_trace", "k_shade", "k_accumulate")):
            insts += float(r["Counter_Value"])
if insts > 0:
    res["valu_wave_insts_per_frame"] = insts / 10.0
try:
    res["bench_under_rocprof"] = json.loads(open(out + "/bench_under_rocprof.json").read().strip().splitlines()[-1])
except Exception as e:
    res["bench_under_rocprof"] = str(e)
# traversal launches together (what bench.py's roofline object times): call-weighted mean over the trace kernels
tk = [k for k in res.get("kernel_calls", {}) if k.startswith("k_trace") or k == "k_extend"]
if tk:
    calls = sum(res["kernel_calls"][k] for k in tk)
    res["trace_launch_avg_us"] = sum(res["kernel_avg_us"][k] * res["kernel_calls"][k] for k in tk) / calls
    hb = res["hbm_traffic_bytes_per_launch"]
    if all(k in hb and hb[k]["bytes_corrected"] is not None for k in tk):
        res["trace_launch_hbm_bytes"] = sum(hb[k]["bytes_corrected"] * res["kernel_calls"][k] for k in tk) / calls
json.dump(res, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "bench_under_rocprof"}, indent=1))
