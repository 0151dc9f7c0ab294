"""Pass shape grid on one GPU: frames_in_flight x frame_batch at a given run length (wall time of draw + wait, best of 3)."""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metal_raytracing_amd as mrt
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=240); ap.add_argument("--warmup", type=int, default=24)
ap.add_argument("--lanes", default="3,4,6,12"); ap.add_argument("--batches", default="4,8,16,32"); ap.add_argument("--bounces", type=int, default=3)
a = ap.parse_args()
w, h = 1920, 1080
scene = mrt.DragonScene((w, h))
for fb in [int(x) for x in a.batches.split(",")]:
    for fl in [int(x) for x in a.lanes.split(",")]:
        r = mrt.Renderer((w, h), scene, seed=1, max_bounces=a.bounces)
        r.set_option("frame_batch", fb); r.set_option("frames_in_flight", fl)
        best = 0.0; reps = []
        for rep in range(3):
            r.draw(a.warmup); r.wait(); r.reset_stats()
            t0 = time.perf_counter(); r.draw(a.steps); r.wait(); dt = time.perf_counter() - t0
            st = r.stats
            reps.append(round((st.closest_rays + st.shadow_rays) / dt / 1e6)); best = max(best, reps[-1])
        print(f"steps {a.steps} frame_batch {fb:2d} lanes {fl:2d} (used {int(r.get_option('lanes_used'))}, {r.get_option('lane_bytes') * r.get_option('lanes_used') / 2**30:6.1f} GiB): {best:8.1f} Mrays/s  reps {reps}", flush=True)
        r.close()
