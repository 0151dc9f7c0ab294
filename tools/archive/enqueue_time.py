"""How long does the host take to queue a 20-frame draw, and how long does the GPU take to run it?"""
import sys, time
sys.path.insert(0, ".")
import metal_raytracing_amd as mrt
sc = mrt.DragonScene((1920, 1080))
r = mrt.Renderer((1920, 1080), sc)
for k, v in (kv.split("=") for kv in sys.argv[1:]): r.set_option(k, float(v))
r.draw(5, wait=True)
for rep in range(5):
    t0 = time.perf_counter(); r.draw(20); t1 = time.perf_counter(); r.wait(); t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0):.3f} ms   until done {1e3 * (t2 - t0):.3f} ms   device {r.stats.ms_gpu_last:.3f} ms", flush=True)
