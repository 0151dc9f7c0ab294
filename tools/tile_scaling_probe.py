"""What one rank of an N-GPU tile-sharded run does, measured on ONE GPU: rank 0 of world N renders its 1/N of the 8x8 tiles for
`steps` frames (the collective is not part of this probe).  Prints wall time, the aggregate rate N ranks would reach if all took
this long, and the efficiency against the 1-GPU run — for several pass shapes (frame_batch), or for the pass size the sharded
renderers themselves choose (`--batches auto`: distributed.py / group.hip: min(32, 8 x world), at most a third of the run).

    python tools/tile_scaling_probe.py --scene garden --width 3840 --height 2160 --batches auto --steps 20      # BASELINE config C4
    python tools/tile_scaling_probe.py --scene dragon4 --batches auto --steps 16                              # BASELINE config C5 (spp 16)
"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metal_raytracing_amd as mrt

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--worlds", default="1,2,4,8"); ap.add_argument("--batches", default="1,2,4,8,16,32")
ap.add_argument("--scene", default="dragon", choices=sorted(mrt.SCENES)); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--bounces", type=int, default=3)
ap.add_argument("--opt", action="append", default=[], help="renderer option key=value (repeatable)")
ap.add_argument("--sopt", action="append", default=[], help="scene option key=value (repeatable), e.g. instancing=1")
ap.add_argument("--all-ranks", action="store_true", help="time EVERY rank r of N (not rank 0 alone): the draw of a real N-GPU run waits for the slowest; prints max / mean over the ranks")
a = ap.parse_args()
w, h = a.width, a.height
scene = mrt.SCENES[a.scene]((w, h))
sopts = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in a.sopt}
ctx = mrt.Context(0)
base = None
print(f"# {a.scene} {w}x{h}, {a.bounces} bounces, {a.steps} timed frames after {a.warmup}; rank 0 of N on one GPU, no collective; best of 3", flush=True)
for world in [int(x) for x in a.worlds.split(",")]:
    batches = ["auto"] if a.batches == "auto" else [int(x) for x in a.batches.split(",")]
    for fb in batches:
      per_rank = []
      for rank in (range(world) if (a.all_ranks and world > 1) else [0]):
        r = mrt.Renderer((w, h), scene, ctx=ctx, seed=1, max_bounces=a.bounces, scene_options=sopts)
        if world > 1: r.set_shard(rank, world)
        if fb == "auto":
            d = int(r.get_option("frame_batch"))
            fb_used = d if world == 1 else min(min(32, d * world), max(d, (a.warmup + a.steps) // 3))      # distributed.py ShardedRenderer
        else:
            fb_used = fb
        r.set_option("frame_batch", fb_used)
        for kv in a.opt:
            k, v = kv.split("="); r.set_option(k, float(v))
        best, best_dt = None, None
        for rep in range(3):
            r.draw(a.warmup); r.wait(); r.reset_stats()
            t0 = time.perf_counter(); r.draw(a.steps); r.wait(); dt = time.perf_counter() - t0
            st = r.stats
            rate = (st.closest_rays + st.shadow_rays) / dt / 1e6
            if best is None or rate > best: best, best_dt, best_rays = rate, dt, st.closest_rays + st.shadow_rays
        if world == 1: base = max(base or 0.0, best)
        if base is None: base = float('nan')
        per_rank.append((rank, best_dt, best, best_rays))
        if not (a.all_ranks and world > 1):
            print(f"world {world} frame_batch {fb_used:2d}: rank time {best_dt * 1e3:7.2f} ms  rank rate {best:8.1f} Mrays/s  x{world} = {best * world:8.1f}  efficiency vs best 1-GPU {best * world / base:5.2f}", flush=True)
        r.close()
      if a.all_ranks and world > 1:
        ts = [x[1] for x in per_rank]; rays = sum(x[3] for x in per_rank)
        print(f"world {world} frame_batch {fb_used:2d}: rank times " + " ".join(f"{t * 1e3:.2f}" for t in ts) + f" ms   max {max(ts) * 1e3:.2f}  mean {sum(ts) / len(ts) * 1e3:.2f}  max/mean {max(ts) / (sum(ts) / len(ts)):.3f}"
              f"   all ranks' rays / slowest rank's time = {rays / max(ts) / 1e6:8.1f} Mrays/s  efficiency vs best 1-GPU {rays / max(ts) / 1e6 / base:5.2f}", flush=True)
