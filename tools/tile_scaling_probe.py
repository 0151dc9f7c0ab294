"""What one rank of an N-GPU tile-sharded run does, measured on ONE GPU: rank 0 of world N renders its 1/N of the 8x8 tiles for
`steps` frames (the collective is not part of this probe).  Prints wall time, the aggregate rate N ranks would reach if all took
this long, and the efficiency against the 1-GPU run — for several pass shapes (frame_batch)."""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metal_raytracing_amd as mrt

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--worlds", default="1,2,4,8"); ap.add_argument("--batches", default="1,2,4,8,16,32")
ap.add_argument("--opt", action="append", default=[], help="renderer option key=value (repeatable), e.g. flow=1")
a = ap.parse_args()
w, h = 1920, 1080
scene = mrt.DragonScene((w, h))
base = None
for world in [int(x) for x in a.worlds.split(",")]:
    for fb in [int(x) for x in a.batches.split(",")]:
        r = mrt.Renderer((w, h), scene, seed=1)
        if world > 1: r.set_shard(0, world)
        r.set_option("frame_batch", fb)
        for kv in a.opt:
            k, v = kv.split("="); r.set_option(k, float(v))
        best = None
        for rep in range(3):
            r.draw(a.warmup); r.wait(); r.reset_stats()
            t0 = time.perf_counter(); r.draw(a.steps); r.wait(); dt = time.perf_counter() - t0
            st = r.stats
            rate = (st.closest_rays + st.shadow_rays) / dt / 1e6
            best = max(best or 0.0, rate)
        if world == 1: base = max(base or 0.0, best)
        if base is None: base = float('nan')
        print(f"world {world} frame_batch {fb:2d}: rank rate {best:8.1f} Mrays/s  x{world} = {best * world:8.1f}  efficiency vs best 1-GPU {best * world / base:5.2f}", flush=True)
        r.close()
