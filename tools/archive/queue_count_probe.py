"""Does the number of HIP streams IN USE on the device change how fast a kernel runs?  
One renderer at the default (6 lanes + the context's stream = 7 streams in use); then k more streams are put to use (a tiny torch kernel on each, kept alive) and the
same draws are timed again: one frame alone (device time, one lane) and 240 frames (wall)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES", "16"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import metal_raytracing_amd as mrt
w, h = 1920, 1080
scene = mrt.DragonScene((w, h))
r = mrt.Renderer((w, h), scene, seed=1)
r.draw(48, wait=True)
q = mrt.Renderer((w, h), scene, ctx=r.ctx, seed=1)
q.set_option("frames_in_flight", 1); q.set_option("frame_batch", 1); q.draw(4, wait=True)
def measure(tag):
    ts = []
    for i in range(16): q.draw(1, wait=True); ts.append(q.stats.ms_gpu_last)
    r.draw(24, wait=True); t0 = time.perf_counter(); r.draw(240, wait=True); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); r.draw(20, wait=True); dt20 = time.perf_counter() - t1
    print(f"{tag}: one frame alone {np.median(ts):.3f} ms; 240 frames {dt / 240 * 1e3:.4f} ms/frame; 20 frames {dt20 / 20 * 1e3:.4f} ms/frame", flush=True)
measure("streams in use: context + 6 lanes + 1 lane of the second renderer = 8")
extra = []
x = torch.zeros(1024, device="cuda")
for k in range(1, 7):
    s = torch.cuda.Stream(); extra.append(s)
    with torch.cuda.stream(s): x.add_(1.0)
    torch.cuda.synchronize()
    measure(f"+ {k} more stream(s) used once (idle now) = {8 + k}")
