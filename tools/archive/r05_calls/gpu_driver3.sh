#!/bin/bash
# the driver's command three times on one box (profiles/ already holds the hash-gated evidence of this tree)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_final; mkdir -p $O; cd $R
for k in 1 2 3; do timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_command_$k.json 2> $O/bench_driver_$k.err; echo "run $k rc=$?"; done
python3 - <<'PY'
import json
for k in (1, 2, 3):
    d = json.loads([l for l in open(f"gpurun_out/r05_final/bench_driver_command_{k}.json") if l.startswith("{")][-1]); r = d["roofline"]
    print(k, d["value"], "Mrays/s", d["ms_per_step"], "ms/step | frac", r["frac"], "avg", r["avg_launch_ms"], "rocprof", r.get("avg_launch_ms_rocprof_serialised_pass"), "| whole chip", r.get("whole_chip", {}).get("frac"), "| ms/frame", d.get("ms_per_frame"), "| strict", d["strict_primary_plus_shadow"]["value"])
PY
