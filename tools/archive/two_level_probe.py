"""Iterations of the stream traversal per ray, flattened against two-level, on dragon x 4: primary rays and bounce-like rays.
usage: tools/two_level_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
sc = mrt.InstancedDragonScene((w, h))
ctx = mrt.Context(0)
cam = sc.camera
ys, xs = np.mgrid[0:h:2, 0:w:2]
px = (xs.ravel() + 0.5) / w * 2 - 1; py = (ys.ravel() + 0.5) / h * 2 - 1
pos = np.array(cam.position.tolist()); right = np.array(cam.right.tolist()); up = np.array(cam.up.tolist()); fwd = np.array(cam.forward.tolist())
d = px[:, None] * right + py[:, None] * up + fwd; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((len(d), 8), np.float32); rays[:, 0:3] = pos; rays[:, 4:7] = d; rays[:, 7] = np.inf
flat = mrt.DeviceScene(ctx, sc, {})
hit = flat.intersect_closest(rays); ok = hit["type"] == 1
P = pos + d[ok] * hit["distance"][ok, None]
rng = np.random.default_rng(3)
nd = rng.normal(size=P.shape); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
br = np.zeros((len(P), 8), np.float32); br[:, 0:3] = P - d[ok] * 1e-3; br[:, 4:7] = nd; br[:, 7] = np.inf
sh = br.copy(); sh[:, 7] = 3.0
for name, opts in (("flattened", {}), ("two-level", {"instancing": 1})):
    ds = flat if not opts else mrt.DeviceScene(ctx, sc, opts)
    for rname, rr, anyh in (("primary", rays, False), ("bounce-like closest", br, False), ("shadow-like any (tmax 3)", sh, True)):
        st = ds.stream_stats(rr, any_hit=anyh, per_wave=1024).astype(np.int64)
        it, live, nodes, tris, n = st[:, 0].sum(), st[:, 1].sum(), st[:, 2].sum(), st[:, 3].sum(), st[:, 7].sum()
        if os.environ.get("MRT_STATS_BOTH"): print(f"      (diagnostics build) per ray: iterations doing a triangle and a node {st[:, 4].sum() / n:.2f}, instance entries {st[:, 5].sum() / n:.2f}")
        print(f"{name:10s} {rname:26s}: {n} rays  wave iterations per 64 rays {64 * it / n:.1f}  live lanes {live / it:.1f}  lane-iterations per ray {live / n:.2f}  node visits {nodes / n:.2f}  triangle / instance steps {tris / n:.2f}", flush=True)
