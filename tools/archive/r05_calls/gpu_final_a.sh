#!/bin/bash
# round 5, final evidence A: the suite, the driver's command and the default bench line, the rocprofv3 evidence (tools/collect_profiles.sh r05)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_final; mkdir -p $O; cd $R
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1; echo "collect rc=$?"; tail -25 $O/collect.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver.err; echo "bench driver rc=$?"
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python3 - <<'PY'
import json
for n in ("bench_driver_command", "bench_default"):
    d = json.loads([l for l in open(f"gpurun_out/r05_final/{n}.json") if l.startswith("{")][-1]); r = d["roofline"]
    print(n, d["value"], "Mrays/s", d["ms_per_step"], "ms/step | roofline frac", r["frac"], "avg", r["avg_launch_ms"], "ms rocprof", r.get("avg_launch_ms_rocprof_serialised_pass"), "traffic", r["traffic"], "| whole chip", r.get("whole_chip", {}).get("frac"), "| overlap", r["under_overlap"]["frac"],
          "| ms/frame", d.get("ms_per_frame"), d["latency"].get("ms_per_frame_as_one_pass"), "3-in-flight", d["latency"]["reference_like_3_in_flight_ms_per_frame"], "| strict", d["strict_primary_plus_shadow"]["value"], "| cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "| parity", d["parity"])
PY
