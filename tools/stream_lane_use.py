"""How full is an iteration of the stream walk?  Bounce-like rays of DragonScene (one from every primary hit, cosine-like about the view-facing direction) through
mrt_debug stream_stats: wave-iterations per 64 rays, and the share of the 64 lanes that are live / test a node / test a triangle in an average iteration.
usage: tools/stream_lane_use.py [per_wave ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
sc = mrt.DragonScene((w, h)); ctx = mrt.Context(0); cam = sc.camera
ty, tx = np.mgrid[0:h // 8:2, 0:w // 8:2]; oy, ox = np.mgrid[0:8, 0:8]
ys = (ty.ravel()[:, None] * 8 + oy.ravel()[None, :]).ravel(); xs = (tx.ravel()[:, None] * 8 + ox.ravel()[None, :]).ravel()
px = (xs + 0.5) / w * 2 - 1; py = (ys + 0.5) / h * 2 - 1
pos = np.array(cam.position.tolist()); right = np.array(cam.right.tolist()); up = np.array(cam.up.tolist()); fwd = np.array(cam.forward.tolist())
d = px[:, None] * right + py[:, None] * up + fwd; d /= np.linalg.norm(d, axis=1, keepdims=True)
prim = np.zeros((len(d), 8), np.float32); prim[:, 0:3] = pos; prim[:, 4:7] = d; prim[:, 7] = np.inf
ds = mrt.DeviceScene(ctx, sc, {})
hit = ds.intersect_closest(prim); ok = hit["type"] == 1
P = pos + d[ok] * hit["distance"][ok, None]
rng = np.random.default_rng(3)
u = rng.normal(size=P.shape); u /= np.linalg.norm(u, axis=1, keepdims=True)
nd = -d[ok] + u; nd /= np.maximum(np.linalg.norm(nd, axis=1, keepdims=True), 1e-6)
rays = np.zeros((len(P), 8), np.float32); rays[:, 0:3] = P - d[ok] * 1e-3; rays[:, 4:7] = nd; rays[:, 7] = np.inf
L = np.array([0, 1.98, 0]) + rng.uniform(-0.25, 0.25, P.shape) * np.array([1, 0, 1])
dl = L - rays[:, 0:3]; dist = np.linalg.norm(dl, axis=1); dl /= dist[:, None]
sh = rays.copy(); sh[:, 4:7] = dl; sh[:, 7] = dist - 1e-3
for pw in [int(a) for a in sys.argv[1:]] or [256, 1024]:
    for name, rr, anyh in (("bounce", rays, False), ("shadow", sh, True)):
        st = ds.stream_stats(np.ascontiguousarray(rr), any_hit=anyh, per_wave=pw).astype(np.int64)
        it, live, node, tri, refills, rl = [int(st[:, k].sum()) for k in range(6)]
        n = len(rr)
        print(f"{name} rays {n}, {pw} per wave: {64 * it / n:6.2f} wave-iterations per 64 rays; lanes live {100 * live / (64 * it):5.1f} %, testing a node {100 * node / (64 * it):5.1f} %, a triangle {100 * tri / (64 * it):5.1f} %; "
              f"lane-iterations per ray {live / n:5.2f} (node {node / n:5.2f}, triangle {tri / n:5.2f}); refills per 64 rays {64 * refills / n:5.2f} ({rl / max(1, refills):4.1f} lanes each)", flush=True)
