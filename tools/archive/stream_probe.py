"""Diagnostics: where do the lanes of the wide stream traversal go?  (live / node / triangle lanes per iteration)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import metal_raytracing_amd as m
from trav_stats import primary_rays
w, h = 1920, 1080
import json
opts = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
print("scene options", opts)
sc = m.DragonScene((w, h)); ctx = m.Context(0); ds = m.DeviceScene(ctx, sc, opts)
rays = primary_rays(w, h)
hit = ds.intersect_closest(rays); ok = hit["type"] == 1
P = rays[ok, 0:3] + rays[ok, 4:7] * hit["distance"][ok, None]
rng = np.random.default_rng(0)
d = rng.normal(size=P.shape).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True); d[:, 1] = np.abs(d[:, 1])
n = (len(P) // 64) * 64
r2 = np.zeros((n, 8), np.float32); r2[:, 0:3] = P[:n] + np.array([0, 2e-3, 0], np.float32); r2[:, 4:7] = d[:n]; r2[:, 7] = np.inf
L = np.array([0, 1.98, 0], np.float32) + rng.uniform(-0.25, 0.25, (n, 3)).astype(np.float32) * np.array([1, 0, 1], np.float32)
dl = L - r2[:, 0:3]; dist = np.linalg.norm(dl, axis=1); dl /= dist[:, None]
r3 = r2.copy(); r3[:, 4:7] = dl; r3[:, 7] = dist - 1e-3
for name, rr, anyh in (("primary", rays, False), ("diffuse", r2, False), ("shadow", r3, True)):
    for pw in (256,):
        st = ds.stream_stats(rr, any_hit=anyh, per_wave=pw).astype(np.float64)
        it, live, node, tri, rf, rfl = (st[:, k].sum() for k in range(6))
        print(f"{name:8s} per_wave={pw:5d}: wave-iterations {it:9.0f} ({it*64/len(rr):6.1f} lane-slots/ray)  live {live/it/64*100:5.1f}%  node lanes {node/it/64*100:5.1f}%  tri lanes {tri/it/64*100:5.1f}%  refills/wave {rf/len(st):5.1f}  lanes/refill {rfl/max(rf,1):5.1f}", flush=True)
