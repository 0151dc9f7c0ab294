"""Would binning rays per instance pay on two-level scenes?  (the structural candidate of DESIGN.md §9.4)

Today a two-level scene is walked in ONE loop: a lane that reaches an instance box changes into object space, walks the BLAS and comes back (traverse_wide_stream<TWO_LEVEL>);
its lanes are out of step — node / triangle / level change — and dragon x 4 costs 10 lane-iterations per ray against 6.6 flattened.  The candidate: a TLAS pass that tests
small instances (<= 8 triangles: walls, floor) in place and QUEUES every (ray, instance) pair whose box the ray enters, then one pass per queue in object space by waves of one
kind (the flattened walk), results merged by min (t, id).  This probe prices it with the kernels that exist, on bounce-like rays of dragon x 4:

  in-loop      stream_stats of the two-level scene                                               (what runs today)
  flattened    stream_stats of the flattened scene                                               (the bar: what one flat tree costs)
  TLAS pass    stream_stats of a two-level scene holding only the small instances                 (lower bound: the large instances' boxes add node tests, no iterations)
  BLAS passes  per large instance: the rays that enter its world box (before the closest small-instance hit), taken into object space on the host, walked through a
               flattened scene of that mesh alone; (a) every pair starts with the TLAS pass's bound, (b) a second round starts with the bound the first left

and reports wave-iterations per 64 rays.  usage: tools/two_level_binning_probe.py [--rays bounce|primary]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt

ap = argparse.ArgumentParser(); ap.add_argument("--rays", default="bounce", choices=["bounce", "primary"]); a = ap.parse_args()
w, h = 1920, 1080
sc = mrt.InstancedDragonScene((w, h))
ctx = mrt.Context(0)
cam = sc.camera
ty, tx = np.mgrid[0:h // 8:2, 0:w // 8:2]                      # every fourth 8x8 tile, pixels in tile order as the renderer's slots are
oy, ox = np.mgrid[0:8, 0:8]
ys = (ty.ravel()[:, None] * 8 + oy.ravel()[None, :]).ravel(); xs = (tx.ravel()[:, None] * 8 + ox.ravel()[None, :]).ravel()
px = (xs + 0.5) / w * 2 - 1; py = (ys + 0.5) / h * 2 - 1
pos = np.array(cam.position.tolist()); right = np.array(cam.right.tolist()); up = np.array(cam.up.tolist()); fwd = np.array(cam.forward.tolist())
d = px[:, None] * right + py[:, None] * up + fwd; d /= np.linalg.norm(d, axis=1, keepdims=True)
prim = np.zeros((len(d), 8), np.float32); prim[:, 0:3] = pos; prim[:, 4:7] = d; prim[:, 7] = np.inf
flat = mrt.DeviceScene(ctx, sc, {})
two = mrt.DeviceScene(ctx, sc, {"instancing": 1})
if a.rays == "primary":
    rays = prim
else:
    hit = flat.intersect_closest(prim); ok = hit["type"] == 1
    P = pos + d[ok] * hit["distance"][ok, None]
    rng = np.random.default_rng(3)
    u = rng.normal(size=P.shape); u /= np.linalg.norm(u, axis=1, keepdims=True)
    nd = -d[ok] + u; nd /= np.maximum(np.linalg.norm(nd, axis=1, keepdims=True), 1e-6)      # cosine-like about the view-facing direction
    rays = np.zeros((len(P), 8), np.float32); rays[:, 0:3] = P - d[ok] * 1e-3; rays[:, 4:7] = nd; rays[:, 7] = np.inf
n = len(rays)
print(f"{a.rays} rays: {n}")


def iters(ds, rr, per_wave=256):
    if len(rr) == 0: return 0, 0
    st = ds.stream_stats(np.ascontiguousarray(rr), any_hit=False, per_wave=per_wave).astype(np.int64)
    return int(st[:, 0].sum()), int(st[:, 1].sum())


it_two, live_two = iters(two, rays); it_flat, live_flat = iters(flat, rays)
print(f"in-loop two-level : {64 * it_two / n:7.2f} wave-iterations per 64 rays   ({live_two / n:5.2f} lane-iterations per ray)")
print(f"flattened         : {64 * it_flat / n:7.2f} wave-iterations per 64 rays   ({live_flat / n:5.2f} lane-iterations per ray)")

# ---- the instances
meshes = sc.meshes
small = [m for m in meshes if m.triangleCount <= 8]
large = [m for m in meshes if m.triangleCount > 8]
print(f"instances: {len(meshes)} ({len(small)} of <= 8 triangles tested in place, {len(large)} queued)")


class Only(mrt.Scene):
    def __init__(self, size, ms, identity=False):
        super().__init__(size)
        self.models = []
        self._ms = ms
    @property
    def meshes(self):
        return self._ms


class _M:      # a mesh under the identity (a BLAS)
    def __init__(self, m):
        self.modelName, self.positions, self.normals, self.submeshes = m.modelName, m.positions, m.normals, m.submeshes
        self.transform = np.eye(4, dtype=np.float32)
    @property
    def triangleCount(self): return sum(s.triangleCount for s in self.submeshes)


# TLAS pass: the small instances in place
tl = Only((w, h), small); tl.lights = sc.lights
tl_ds = mrt.DeviceScene(ctx, tl, {"instancing": 1})
it_tlas, _ = iters(tl_ds, rays)
hs = tl_ds.intersect_closest(rays)
bound = np.where(hs["type"] == 1, hs["distance"], np.inf).astype(np.float32)      # the closest small-instance hit: every queued pair starts with it
print(f"TLAS pass (small instances in place): {64 * it_tlas / n:7.2f} wave-iterations per 64 rays")

# BLAS passes
o, dd = rays[:, 0:3].astype(np.float64), rays[:, 4:7].astype(np.float64)
pairs_total, it_blas_a, it_blas_b = 0, 0, 0
best = bound.copy()
blas_cache = {}
entries = []
for m in large:
    M = m.transform.astype(np.float64).T                       # (4,4) [col][row] -> row-major 4x4
    lo, hi = m.positions.min(0).astype(np.float64), m.positions.max(0).astype(np.float64)
    Minv = np.linalg.inv(M)
    oo = (Minv[:3, :3] @ o.T).T + Minv[:3, 3]; od = (Minv[:3, :3] @ dd.T).T          # direction NOT renormalised: t stays the world distance
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / np.where(np.abs(od) < 1e-20, 1e-20, od)
        t0 = (lo - oo) * inv; t1 = (hi - oo) * inv
    tn = np.maximum(np.minimum(t0, t1).max(1), 0.0); tf = np.maximum(t0, t1).min(1)
    enters = (tn <= tf) & (tn < bound)
    key = id(m.positions)
    if key not in blas_cache:
        b = Only((w, h), [_M(m)]); b.lights = sc.lights
        blas_cache[key] = mrt.DeviceScene(ctx, b, {})
    entries.append((m, enters, tn, oo.astype(np.float32), od.astype(np.float32), blas_cache[key]))
    pairs_total += int(enters.sum())
print(f"(ray, instance) pairs queued: {pairs_total} = {pairs_total / n:.2f} per ray")
# (a) every pair starts with the TLAS pass's bound
for m, enters, tn, oo, od, ds in entries:
    rr = np.zeros((int(enters.sum()), 8), np.float32); rr[:, 0:3] = oo[enters]; rr[:, 4:7] = od[enters]; rr[:, 7] = bound[enters]
    it, _ = iters(ds, rr); it_blas_a += it
    hh = ds.intersect_closest(rr) if len(rr) else None
    if hh is not None:
        t = np.where(hh["type"] == 1, hh["distance"], np.inf).astype(np.float32)
        idx = np.nonzero(enters)[0]; best[idx] = np.minimum(best[idx], t)
print(f"BLAS passes (a) one round, TLAS bound   : {64 * it_blas_a / n:7.2f} wave-iterations per 64 rays (of the ORIGINAL ray count)")
# (b) two rounds: each ray's nearest-entry instance first, the others afterwards with the bound that round left
first_t = np.full(n, np.inf); first_k = np.full(n, -1)
for k, (m, enters, tn, *_) in enumerate(entries):
    better = enters & (tn < first_t); first_t[better] = tn[better]; first_k[better] = k
bound1 = bound.copy(); it_b1 = it_b2 = 0
for k, (m, enters, tn, oo, od, ds) in enumerate(entries):
    sel = enters & (first_k == k)
    rr = np.zeros((int(sel.sum()), 8), np.float32); rr[:, 0:3] = oo[sel]; rr[:, 4:7] = od[sel]; rr[:, 7] = bound[sel]
    it, _ = iters(ds, rr); it_b1 += it
    if len(rr):
        hh = ds.intersect_closest(rr); t = np.where(hh["type"] == 1, hh["distance"], np.inf).astype(np.float32)
        idx = np.nonzero(sel)[0]; bound1[idx] = np.minimum(bound1[idx], t)
pairs2 = 0
for k, (m, enters, tn, oo, od, ds) in enumerate(entries):
    sel = enters & (first_k != k) & (tn < bound1)
    rr = np.zeros((int(sel.sum()), 8), np.float32); rr[:, 0:3] = oo[sel]; rr[:, 4:7] = od[sel]; rr[:, 7] = bound1[sel]
    it, _ = iters(ds, rr); it_b2 += it; pairs2 += int(sel.sum())
print(f"BLAS passes (b) nearest box first + rest: {64 * (it_b1 + it_b2) / n:7.2f} wave-iterations per 64 rays ({64 * it_b1 / n:.2f} + {64 * it_b2 / n:.2f}; second round {pairs2 / n:.2f} pairs per ray)")
for name, tot in (("binned (a)", it_tlas + it_blas_a), ("binned (b)", it_tlas + it_b1 + it_b2)):
    print(f"{name}: {64 * tot / n:7.2f} wave-iterations per 64 rays = {tot / it_two:5.2f} x in-loop two-level, {tot / it_flat:5.2f} x flattened   (+ a queue write and read of 32-48 B per pair, + the merge)")
