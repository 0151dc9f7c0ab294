"""Is the frame rate bound by the host?  Times the enqueue of N frames (Renderer.draw without waiting) against their completion."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metal_raytracing_amd as m
w, h = 1920, 1080
sc = m.DragonScene((w, h))
r = m.Renderer((w, h), sc)
r.draw(24, wait=True)
for n in (96, 96, 192):
    t0 = time.perf_counter(); r.draw(n, wait=False); t1 = time.perf_counter(); r.wait(); t2 = time.perf_counter()
    print(f"{n} frames: enqueue {1e3 * (t1 - t0):7.2f} ms ({1e6 * (t1 - t0) / n:6.1f} us/frame), complete {1e3 * (t2 - t0):7.2f} ms ({1e3 * (t2 - t0) / n:.4f} ms/frame)", flush=True)
