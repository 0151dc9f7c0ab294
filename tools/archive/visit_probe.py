"""Diagnostics: how many 8-wide node visits are wasted?  Per bounce-like ray (traverse_wide<STATS>): node visits, visits in which no child was
hit ("empty"), visits whose own grid box already lay beyond the best hit when the node was popped ("stale": a distance kept with the stack entry
would have skipped them), triangle tests.  usage: tools/visit_probe.py ['{"wide_collapse": 0}' ...]"""
import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
sc = mrt.DragonScene((w, h))
ctx = mrt.Context(0)
cam = sc.camera
ys, xs = np.mgrid[0:h:3, 0:w:3]
px = (xs.ravel() + 0.5) / w * 2 - 1; py = (ys.ravel() + 0.5) / h * 2 - 1
pos = np.array(cam.position.tolist()); right = np.array(cam.right.tolist()); up = np.array(cam.up.tolist()); fwd = np.array(cam.forward.tolist())
d = px[:, None] * right + py[:, None] * up + fwd; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((len(d), 8), np.float32); rays[:, 0:3] = pos; rays[:, 4:7] = d; rays[:, 7] = np.inf
for opts in [json.loads(a) for a in sys.argv[1:]] or [{}]:
    ds = mrt.DeviceScene(ctx, sc, opts)
    hit = ds.intersect_closest(rays)
    ok = hit["type"] == 1
    P = pos + d[ok] * hit["distance"][ok, None]
    rng = np.random.default_rng(3)
    nd = rng.normal(size=P.shape); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    br = np.zeros((len(P), 8), np.float32); br[:, 0:3] = P - d[ok] * 1e-3; br[:, 4:7] = nd; br[:, 7] = np.inf
    for name, rr, anyh in (("bounce-like closest", br, False), ("bounce-like any (tmax 3)", np.concatenate([br[:, :7], np.full((len(br), 1), 3.0, np.float32)], 1), True)):
        st = ds.traversal_stats(rr, any_hit=anyh).astype(np.int64)
        nodes, tris, empty, stale = st[:, 0], st[:, 2], st[:, 7] & 0xFFFF, st[:, 7] >> 16
        print(f"{opts} {name}: {len(rr)} rays  node visits {nodes.mean():.2f}  empty {empty.mean():.2f} ({100*empty.sum()/nodes.sum():.0f}%)  stale {stale.mean():.2f} ({100*stale.sum()/nodes.sum():.0f}%)  "
              f"triangle tests {tris.mean():.2f}  hit {100*(st[:,3]!=0xFFFFFFFF).mean():.0f}%", flush=True)
    ds.close()
