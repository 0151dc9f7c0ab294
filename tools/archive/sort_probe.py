"""Would sorting the bounce rays by direction octant pay?  Bounce-like rays (primary hit points of the dragon scene, cosine-like directions about the view-facing normal) walked by the
stream traversal in queue order, sorted by octant inside blocks of 256 (what k_shade's compaction could do locally), and sorted globally: wave iterations per 64 rays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
sc = mrt.DragonScene((w, h))
ctx = mrt.Context(0)
cam = sc.camera
# pixels in 8x8 tile order, as the renderer's slots are
ty, tx = np.mgrid[0:h // 8, 0:w // 8]
oy, ox = np.mgrid[0:8, 0:8]
ys = (ty.ravel()[:, None] * 8 + oy.ravel()[None, :]).ravel(); xs = (tx.ravel()[:, None] * 8 + ox.ravel()[None, :]).ravel()
px = (xs + 0.5) / w * 2 - 1; py = 1 - (ys + 0.5) / h * 2
pos = np.array(cam.position.tolist()); right = np.array(cam.right.tolist()); up = np.array(cam.up.tolist()); fwd = np.array(cam.forward.tolist())
d = px[:, None] * right * (w / h) * 0.41 + py[:, None] * up * 0.41 + fwd; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((len(d), 8), np.float32); rays[:, 0:3] = pos; rays[:, 4:7] = d; rays[:, 7] = np.inf
ds = mrt.DeviceScene(ctx, sc, {})
hit = ds.intersect_closest(rays); ok = hit["type"] == 1
print("primary hits", ok.mean())
P = pos + d[ok] * hit["distance"][ok, None]
rng = np.random.default_rng(3)
u = rng.normal(size=P.shape); u /= np.linalg.norm(u, axis=1, keepdims=True)
nd = -d[ok] + u; nd /= np.maximum(np.linalg.norm(nd, axis=1, keepdims=True), 1e-6)
br = np.zeros((len(P), 8), np.float32); br[:, 0:3] = P - d[ok] * 1e-3; br[:, 4:7] = nd; br[:, 7] = np.inf
octant = (nd[:, 0] < 0) * 1 + (nd[:, 1] < 0) * 2 + (nd[:, 2] < 0) * 4
def run(name, order):
    st = ds.stream_stats(np.ascontiguousarray(br[order]), any_hit=False, per_wave=256).astype(np.int64)
    it, live, n = st[:, 0].sum(), st[:, 1].sum(), st[:, 7].sum()
    print(f"{name:34s}: {n} rays  wave iterations per 64 rays {64 * it / n:6.2f}  live lanes {live / it:5.1f}", flush=True)
n = len(br)
run("queue order", np.arange(n))
blk = np.arange(n) // 256
run("octant-sorted inside blocks of 256", np.lexsort((octant, blk)))
blk = np.arange(n) // 2048
run("octant-sorted inside blocks of 2048", np.lexsort((octant, blk)))
run("octant-sorted globally (stable)", np.argsort(octant, kind="stable"))
