"""Tile groups (renderer option tile_groups) in the small-launch regimes, same process, alternating:
  one frame alone (frames_in_flight = 1, frame_batch = 1): device time of the frame, median of 24
  the reference's regime (three one-frame passes in flight, Renderer.swift:33): wall time per frame over 30 frames
  a rank of eight over the driver's 20 frames (rank 0 of 8, frame_batch as the sharded renderers choose): wall time of the draw
  the driver's command on one GPU (20 frames after 5): Mrays/s"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
scene = mrt.DragonScene((w, h)); ctx = mrt.Context(0)
extra = dict(kv.split("=") for kv in sys.argv[1:])
def mk(groups, **o):
    r = mrt.Renderer((w, h), scene, ctx=ctx, seed=1)
    r.set_option("tile_groups", groups)
    for k, v in {**o, **extra}.items(): r.set_option(k, float(v))
    return r
for rep in range(2):
    for g in (1, 2, 3, 4, 0):
        r = mk(g, frames_in_flight=1, frame_batch=1)
        r.draw(4, wait=True); ts = []
        for i in range(24): r.draw(1, wait=True); ts.append(r.stats.ms_gpu_last)
        print(f"one frame alone, tile_groups {g} (used {int(r.get_option('groups_used'))}): median {np.median(ts):.4f} ms  min {np.min(ts):.4f}", flush=True)
        r.close()
    for g in (1, 2, 0):
        r = mk(g, frames_in_flight=3, frame_batch=1)
        r.draw(6, wait=True); t0 = time.perf_counter(); r.draw(30, wait=True); dt = time.perf_counter() - t0
        print(f"three one-frame passes in flight, tile_groups {g} (used {int(r.get_option('groups_used'))}): {dt * 1e3 / 30:.4f} ms per frame", flush=True)
        r.close()
    for g in (1, 2, 0):
        r = mk(g); r.set_shard(0, 8); r.set_option("frame_batch", 8)
        best = 1e9
        for k in range(3):
            r.draw(5, wait=True); t0 = time.perf_counter(); r.draw(20, wait=True); best = min(best, time.perf_counter() - t0)
        print(f"rank 0 of 8 over 20 frames (frame_batch 8), tile_groups {g} (used {int(r.get_option('groups_used'))}): {best * 1e3:.3f} ms", flush=True)
        r.close()
    for g in (1, 2, 0):
        r = mk(g)
        best = 0
        for k in range(3):
            r.draw(5, wait=True); r.reset_stats(); t0 = time.perf_counter(); r.draw(20, wait=True); dt = time.perf_counter() - t0
            st = r.stats; best = max(best, (st.closest_rays + st.shadow_rays) / dt / 1e6)
        print(f"driver's 20 frames, tile_groups {g} (used {int(r.get_option('groups_used'))}): {best:.1f} Mrays/s", flush=True)
        r.close()
