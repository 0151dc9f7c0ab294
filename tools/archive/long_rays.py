"""Which rays are the stragglers?  Per-ray iteration counts (8-wide node visits + triangle tests, traverse_wide<STATS>) for bounce-like rays:
origins on the visible surfaces (primary hits of a pixel subsample), uniformly random directions."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
sc = mrt.DragonScene((w, h))
r = mrt.Renderer((w, h), sc, seed=1)
ds = r.device_scene
cam = sc.camera
ys, xs = np.mgrid[0:h:3, 0:w:3]
px = (xs.ravel() + 0.5) / w * 2 - 1; py = (ys.ravel() + 0.5) / h * 2 - 1
pos = np.array(cam.position.tolist()); right = np.array(cam.right.tolist()); up = np.array(cam.up.tolist()); fwd = np.array(cam.forward.tolist())
d = px[:, None] * right + py[:, None] * up + fwd; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((len(d), 8), np.float32); rays[:, 0:3] = pos; rays[:, 4:7] = d; rays[:, 7] = np.inf
hit = ds.intersect_closest(rays)
ok = hit["type"] == 1
P = pos + d[ok] * hit["distance"][ok, None]
rng = np.random.default_rng(3)
nd = rng.normal(size=P.shape); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
br = np.zeros((len(P), 8), np.float32); br[:, 0:3] = P - d[ok] * 1e-3; br[:, 4:7] = nd; br[:, 7] = np.inf
for name, rr, anyh in (("primary", rays, False), ("bounce-like closest", br, False), ("bounce-like any (tmax 3)", np.concatenate([br[:, :7], np.full((len(br), 1), 3.0, np.float32)], 1), True)):
    st = ds.traversal_stats(rr, any_hit=anyh).astype(np.int64)
    it = st[:, 0] + st[:, 2]
    q = np.percentile(it, [50, 90, 99, 99.9, 99.99])
    print(f"{name}: {len(it)} rays, iterations mean {it.mean():.1f} p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} p99.99 {q[4]:.0f} max {it.max()}  (nodes mean {st[:,0].mean():.1f}, tris mean {st[:,2].mean():.1f})")
    worst = np.argsort(-it)[:5]
    for k in worst:
        print("    ", it[k], "nodes", st[k, 0], "tris", st[k, 2], "o", np.round(rr[k, 0:3], 3), "d", np.round(rr[k, 4:7], 3), "hit gid", st[k, 3])
