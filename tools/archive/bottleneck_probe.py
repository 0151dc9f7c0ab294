"""Diagnostics: is the rope traversal kernel bound by VALU issue or by the vector-memory path?  Repeats the
box arithmetic (alu_dup) or the node fetch (mem_dup) inside the loop and reports bulk / total time."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import metal_raytracing_amd as m
from trav_stats import primary_rays

def timing(st):
    t0 = st[:, 4].astype(np.int64).reshape(-1, 64)[:, 0]; t1 = st[:, 5].astype(np.int64).reshape(-1, 64).max(1)
    base = t0.min(); end = np.sort(t1 - base) / 100.0
    return end[int(0.95 * len(end))], end[-1]

w, h = 1920, 1080
sc = m.DragonScene((w, h)); ctx = m.Context(0)
ds = m.DeviceScene(ctx, sc, {"wide": 0})
rays = primary_rays(w, h)
hit = ds.intersect_closest(rays); ok = hit["type"] == 1
P = rays[ok, 0:3] + rays[ok, 4:7] * hit["distance"][ok, None]
rng = np.random.default_rng(0)
d = rng.normal(size=P.shape).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True); d[:, 1] = np.abs(d[:, 1])
n = (len(P) // 64) * 64
r2 = np.zeros((n, 8), np.float32); r2[:, 0:3] = P[:n] + np.array([0, 2e-3, 0], np.float32); r2[:, 4:7] = d[:n]; r2[:, 7] = np.inf
for name, rr in (("primary", rays), ("diffuse", r2)):
    for (a, mm) in [(0, 0), (0, 0), (1, 0), (2, 0), (4, 0), (0, 1), (0, 2)]:
        st = ds.traversal_stats(rr, alu_dup=a, mem_dup=mm)
        b, t = timing(st)
        print(f"{name:8s} alu_dup={a} mem_dup={mm}: 95% of waves done at {b:7.1f} us, kernel {t:7.1f} us", flush=True)
