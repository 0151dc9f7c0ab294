#!/bin/bash
# round 5, final evidence on the final tree: the rocprofv3 evidence (tools/collect_profiles.sh r05), then bench.py's own lines WITH that evidence in profiles/ (hash-gated)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_final; mkdir -p $O; cd $R
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1; rc=$?; echo "collect rc=$rc"; [ $rc -eq 0 ] || { tail -20 $O/collect.log; exit 1; }
cp gpurun_out/profiles_r05/r05_* profiles/
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver.err; echo "bench driver rc=$?"
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python3 - <<'PY'
import json
for n in ("bench_driver_command", "bench_default"):
    d = json.loads([l for l in open(f"gpurun_out/r05_final/{n}.json") if l.startswith("{")][-1]); r = d["roofline"]
    print(n, d["value"], "Mrays/s", d["ms_per_step"], "ms/step | roofline frac", r["frac"], "avg", r["avg_launch_ms"], "ms rocprof", r.get("avg_launch_ms_rocprof_serialised_pass"), "traffic", r["traffic"], "| whole chip", r.get("whole_chip", {}).get("frac"), "| overlap", r["under_overlap"]["frac"],
          "| ms/frame", d.get("ms_per_frame"), d["latency"].get("ms_per_frame_as_one_pass"), "3-in-flight", d["latency"]["reference_like_3_in_flight_ms_per_frame"], "| strict", d["strict_primary_plus_shadow"]["value"], "| cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "| parity", d["parity"])
PY
