"""Round-5 probes of the stream walk on DragonScene's bounce-like and shadow rays (tools/stream_lane_use.py's rays), through diagnostics builds of the library:
  MRT_LIB_PATH=.../libmrt_hip_probe1.so (-DMRT_STATS_PROBE=1)  node fetches by tree level: what an LDS copy of the top levels would serve
  MRT_LIB_PATH=.../libmrt_hip_probe2.so (-DMRT_STATS_PROBE=2)  pending triangles per iteration: what pooling triangle tests across lanes could use
usage: tools/stream_level_probe.py 1|2 [per_wave]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
mode = int(sys.argv[1]); pw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
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
st_ = ds.stats
print(f"scene: {st_.triangles} triangles, {st_.bvh_nodes} 8-wide nodes, depth {st_.wide_depth}", flush=True)
for name, rr, anyh in (("bounce", rays, False), ("shadow", sh, True), ("primary (tile order)", prim, False)):
    st = ds.stream_stats(np.ascontiguousarray(rr), any_hit=anyh, per_wave=pw).astype(np.int64)
    it, live, node, tri, a, b, c = [int(st[:, k].sum()) for k in range(7)]
    n = len(rr)
    if mode == 1:
        print(f"{name} rays {n}: node fetches per ray {node / n:5.2f}; of them in levels 0-1 {100 * a / node:5.1f} %, levels 0-2 {100 * b / node:5.1f} %, levels 0-3 {100 * c / node:5.1f} %  (wave-iterations per 64 rays {64 * it / n:5.2f})", flush=True)
    else:
        print(f"{name} rays {n}: {64 * it / n:5.2f} wave-iterations per 64 rays; lanes testing a triangle {100 * tri / (64 * it):5.1f} %, a node {100 * node / (64 * it):5.1f} %; pending triangles per iteration {a / it:6.1f} (per live lane {a / max(1, live):4.2f}); "
              f"lanes with two or more pending {100 * b / (64 * it):5.1f} %; iterations with fewer than 16 triangle lanes {100 * c / it:5.1f} %", flush=True)
