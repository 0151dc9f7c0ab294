"""Does bounding the persistent traversal grid (wave_slots) leave room for the other passes' shade / primary waves?  lanes x wave_slots, 240 steps."""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metal_raytracing_amd as mrt
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=240); ap.add_argument("--warmup", type=int, default=24)
ap.add_argument("--lanes", default="3,4,6"); ap.add_argument("--slots", default="0,5376,3584,2560,1792"); ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
w, h = 1920, 1080
scene = mrt.DragonScene((w, h))
for fl in [int(x) for x in a.lanes.split(",")]:
    for ws in [int(x) for x in a.slots.split(",")]:
        r = mrt.Renderer((w, h), scene, seed=1)
        r.set_option("frame_batch", a.batch); r.set_option("frames_in_flight", fl)
        if ws: r.set_option("wave_slots", ws)
        reps = []
        for rep in range(3):
            r.draw(a.warmup); r.wait(); r.reset_stats()
            t0 = time.perf_counter(); r.draw(a.steps); r.wait(); dt = time.perf_counter() - t0
            st = r.stats
            reps.append(round((st.closest_rays + st.shadow_rays) / dt / 1e6))
        kt = r.kernel_times
        print(f"lanes {fl} batch {a.batch} wave_slots {ws or 'all'}: {max(reps)} Mrays/s {reps}  avg launch ms " + " ".join(f"{k}={ms / max(n, 1):.3f}" for k, (ms, n) in kt.items()), flush=True)
        r.close()
