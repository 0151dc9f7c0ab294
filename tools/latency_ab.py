"""Same-process A/B of renderer option sets in the small-launch regimes, alternating the sets REPS times:
  one frame alone (frames_in_flight = 1, frame_batch = 1): device time of the frame, median of 32
  the reference's regime (three one-frame passes in flight, Renderer.swift:33): wall time per frame over 30 frames
  a rank of eight over the driver's 20 frames (rank 0 of 8): wall time of the draw, best of 3
  the driver's command on one GPU (20 frames after 5): Mrays/s, best of 3
usage: tools/latency_ab.py [reps=N] SET [SET ...]      SET = '-' (defaults) or k=v,k=v"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
scene = mrt.DragonScene((w, h)); ctx = mrt.Context(0)
args = sys.argv[1:]; reps = 3
if args and args[0].startswith("reps="): reps = int(args.pop(0)[5:])
sets = [(a, {} if a == "-" else dict(kv.split("=") for kv in a.split(","))) for a in args]
def mk(o, **base):
    r = mrt.Renderer((w, h), scene, ctx=ctx, seed=1)
    for k, v in {**base, **o}.items(): r.set_option(k, float(v))
    return r
res = {}
def note(regime, name, v): res.setdefault((regime, name), []).append(v)
for rep in range(reps):
    for name, o in sets:
        r = mk(o, frames_in_flight=1, frame_batch=1)
        r.draw(4, wait=True); ts = []
        for i in range(32): r.draw(1, wait=True); ts.append(r.stats.ms_gpu_last)
        note("one frame alone, ms", name, float(np.median(ts))); r.close()
    for name, o in sets:
        r = mk(o, frames_in_flight=3, frame_batch=1)
        r.draw(6, wait=True); t0 = time.perf_counter(); r.draw(30, wait=True); dt = time.perf_counter() - t0
        note("three one-frame passes in flight, ms per frame", name, dt * 1e3 / 30); r.close()
    for name, o in sets:
        r = mk(o); r.set_shard(0, 8); r.set_option("frame_batch", 8)
        best = 1e9
        for k in range(3):
            r.draw(5, wait=True); t0 = time.perf_counter(); r.draw(20, wait=True); best = min(best, time.perf_counter() - t0)
        note("rank 0 of 8 over 20 frames, ms", name, best * 1e3); r.close()
    for name, o in sets:
        r = mk(o); best = 0
        for k in range(3):
            r.draw(5, wait=True); r.reset_stats(); t0 = time.perf_counter(); r.draw(20, wait=True); dt = time.perf_counter() - t0
            st = r.stats; best = max(best, (st.closest_rays + st.shadow_rays) / dt / 1e6)
        note("driver's 20 frames, Mrays/s", name, best); r.close()
    print(f"rep {rep} done", flush=True)
for (regime, name), v in res.items():
    print(f"{regime:48s} [{name:40s}] median {np.median(v):9.4f}   all {' '.join(f'{x:.4f}' for x in v)}", flush=True)
