"""Is the frame driver launch-bound?  Host time to ENQUEUE n frames (Renderer.draw returns after the last launch) against the device time they take."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metal_raytracing_amd as mrt
w, h = 1920, 1080
r = mrt.Renderer((w, h), mrt.DragonScene((w, h)), seed=1)
r.draw(24, wait=True)
for n, fb, fl in ((240, 4, 12), (20, 4, 12), (24, 1, 1), (24, 1, 3)):
    r.set_option("frame_batch", fb); r.set_option("frames_in_flight", fl); r.draw(max(fb, 4), wait=True)
    t0 = time.perf_counter(); r.draw(n); t1 = time.perf_counter(); r.wait(); t2 = time.perf_counter()
    passes = (n + fb - 1) // fb
    print(f"{n} frames, {fb} per pass, {fl} lanes: enqueue {1e3 * (t1 - t0):.2f} ms ({1e6 * (t1 - t0) / passes:.0f} us per pass of 8 launches), until done {1e3 * (t2 - t0):.2f} ms")
r.close()
