"""Randomised self-consistency: the image and the ray counts must not depend on frame_bundle (0 / 1 / 2), halton_table, frame_batch, frames_in_flight, shards or tile groups."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import metal_raytracing_amd as mrt
rng = np.random.default_rng(11)
ctx = mrt.Context(0)
bad = 0
for trial in range(40):
    name = rng.choice(["dragon", "cornell", "garden", "dragon_hostile"])
    w, h = int(rng.integers(17, 400)), int(rng.integers(9, 240))
    bounces = int(rng.integers(1, 5)); frames = int(rng.integers(2, 40))
    sc = mrt.SCENES[name]((w, h))
    ref = None
    for cfg in range(4):
        r = mrt.Renderer((w, h), sc, ctx=ctx, max_bounces=bounces)
        if cfg == 0: o = dict(frame_bundle=0, halton_table=0)
        else: o = dict(frame_bundle=int(rng.integers(1, 3)), halton_table=int(rng.integers(0, 2)), frame_batch=int(rng.integers(1, 33)), frames_in_flight=int(rng.integers(1, 7)))
        for k, v in o.items(): r.set_option(k, v)
        world = int(rng.integers(1, 4)) if cfg >= 2 else 1
        acc = np.zeros((h, w, 4), np.float32); cnt = [0, 0]
        for rank in range(world):
            if world > 1: r.set_shard(rank, world); r.frameIndex = 0
            # two draws: the second continues the accumulation
            a = frames // 2
            r.reset_stats(); r.draw(a, wait=True); r.draw(frames - a, wait=True)
            img = r.accumulation(); acc += img if world > 1 else 0
            if world == 1: acc = img.copy()
            cnt[0] += r.stats.closest_rays; cnt[1] += r.stats.shadow_rays
        r.close()
        if ref is None: ref = (acc.copy(), tuple(cnt))
        else:
            same = np.array_equal(acc[..., :3].view(np.uint32), ref[0][..., :3].view(np.uint32)) and tuple(cnt) == ref[1]
            if not same: bad += 1; print("MISMATCH", name, (w, h), bounces, frames, o, world, tuple(cnt), ref[1], float(np.abs(acc[..., :3] - ref[0][..., :3]).max()), flush=True)
    print(f"trial {trial}: {name} {w}x{h} bounces {bounces} frames {frames} ok", flush=True)
print("mismatches:", bad)
