"""Diagnostics: how much faster do incoherent (diffuse bounce) rays traverse when the queue is ordered by
direction octant and origin cell?  Uses the per-wave timestamps of the stats query kernel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import metal_raytracing_amd as m
from trav_stats import primary_rays

def timing(st):
    t0 = st[:, 4].astype(np.int64).reshape(-1, 64)[:, 0]; t1 = st[:, 5].astype(np.int64).reshape(-1, 64).max(1)
    base = t0.min(); end = np.sort(t1 - base) / 100.0
    wi = st[:, 6].astype(np.int64).reshape(-1, 64).sum(1)
    return end[int(0.95 * len(end))], end[-1], wi.sum()

def morton3(q):
    def spread(x):
        x = x.astype(np.uint64) & 0x3ff
        x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249
        return x
    return (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])

w, h = 1920, 1080
sc = m.DragonScene((w, h)); ctx = m.Context(0)
for wide in (1, 0):
    ds = m.DeviceScene(ctx, sc, {"wide": wide})
    rays = primary_rays(w, h)
    hit = ds.intersect_closest(rays); ok = hit["type"] == 1
    P = rays[ok, 0:3] + rays[ok, 4:7] * hit["distance"][ok, None]
    rng = np.random.default_rng(0)
    d = rng.normal(size=P.shape).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True); d[:, 1] = np.abs(d[:, 1])
    n = (len(P) // 64) * 64
    r2 = np.zeros((n, 8), np.float32); r2[:, 0:3] = P[:n] + np.array([0, 2e-3, 0], np.float32); r2[:, 4:7] = d[:n]; r2[:, 7] = np.inf
    octant = (r2[:, 4] < 0).astype(np.uint64) | ((r2[:, 5] < 0).astype(np.uint64) << 1) | ((r2[:, 6] < 0).astype(np.uint64) << 2)
    lo, hi = r2[:, 0:3].min(0), r2[:, 0:3].max(0)
    for bits in (0, 3, 5, 8):
        if bits == 0:
            order = np.arange(n); name = "queue order"
        else:
            q = np.clip(((r2[:, 0:3] - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
            key = (octant << np.uint64(32)) | morton3(q)
            order = np.argsort(key, kind="stable"); name = f"octant + {bits}-bit/axis origin cell"
        st = ds.traversal_stats(r2[order])
        b, t, wi = timing(st)
        print(f"wide={wide} {name:34s}: 95% waves done {b:7.1f} us, kernel {t:7.1f} us, wave-iterations {wi}", flush=True)
    ds.close()
