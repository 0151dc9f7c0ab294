"""Runs bench.py (no CPU baseline) and prints the headline numbers in one line — for iteration."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "30", "--warmup", "3"] + args, capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    print(f"value={d['value']} Mrays/s  ms/frame={d['ms_per_step']}  trace avg {r['avg_launch_ms']} ms ({r['achieved']} GB/s alg)  build {d['config']['bvh_build_ms']} ms  sah {d['config']['sah_cost']}")
except Exception as e:
    print("bench failed:", e, out.stdout[-2000:], out.stderr[-3000:])
