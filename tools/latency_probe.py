"""Device time of ONE frame alone (frames_in_flight = 1, frame_batch = 1) for a few renderer option sets: median of 24 frames."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
scene = mrt.DragonScene((w, h))
sets = [dict(), dict(persistent=1), dict(persistent=1, persist_chunk=128), dict(persistent=1, persist_chunk=64), dict(persistent=0), dict(megakernel=1)]
for extra in sys.argv[1:]:
    sets.append(dict(kv.split("=") for kv in extra.split(",")))
for o in sets:
    r = mrt.Renderer((w, h), scene, seed=1)
    r.set_option("frames_in_flight", 1); r.set_option("frame_batch", 1)
    for k, v in o.items(): r.set_option(k, float(v))
    r.draw(4, wait=True)
    ts = []
    for i in range(24):
        r.draw(1, wait=True); ts.append(r.stats.ms_gpu_last)
    kt = r.kernel_times
    print(o, f"median {np.median(ts):.4f} ms  min {np.min(ts):.4f}", {k: round(ms / max(n, 1), 4) for k, (ms, n) in kt.items()}, flush=True)
    r.close()
