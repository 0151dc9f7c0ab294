"""Diagnostics: traversal work per ray (node visits, leaf visits, triangle tests) for the primary rays
of the benchmark camera and for diffuse / shadow-like rays, per BVH builder option set."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metal_raytracing_amd as m

def primary_rays(w, h, step=1):
    cam = m.Scene.setupCamera((w, h))
    tx, ty = np.meshgrid(np.arange(0, w // 8, step), np.arange(0, h // 8, step))
    k = np.arange(64)
    xs = (tx.reshape(-1, 1) * 8 + (k & 7)).reshape(-1); ys = (ty.reshape(-1, 1) * 8 + (k >> 3)).reshape(-1)
    u = (xs + 0.5) / w * 2 - 1; v = (ys + 0.5) / h * 2 - 1
    d = np.outer(u, cam.right.tolist()) + np.outer(v, cam.up.tolist()) + np.array(cam.forward.tolist())
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((len(xs), 8), np.float32)
    rays[:, 0:3] = cam.position.tolist(); rays[:, 4:7] = d; rays[:, 7] = np.inf
    return rays

def report(tag, st):
    steps, leaves, tris = (st[:, 0] + st[:, 2]).astype(np.float64), st[:, 1], st[:, 2]
    wave_max = steps.reshape(-1, 64).max(1)
    util = steps.reshape(-1, 64).sum(1) / (64 * np.maximum(wave_max, 1))
    t0 = st[:, 4].astype(np.int64).reshape(-1, 64)[:, 0]; t1 = st[:, 5].astype(np.int64).reshape(-1, 64).max(1)
    base = t0.min(); span = (t1.max() - base) / 100.0
    dur = (t1 - t0) / 100.0
    ts = np.linspace(0, span, 21)[1:]
    occ = [int(((t0 - base) / 100.0 <= x).sum() - ((t1 - base) / 100.0 <= x).sum()) for x in ts]
    wi = st[:, 6].astype(np.int64).reshape(-1, 64)
    inner = wi.sum(1); leafph = np.zeros_like(inner)
    k = int(np.argmax(dur))
    print(f"     slowest wave #{k}: {dur[k]:.1f} us, inner iterations {inner[k]}, leaf phases {leafph[k]}, max lane steps {st[:,0].reshape(-1,64)[k].max()}, sum lane steps {st[:,0].reshape(-1,64)[k].sum()} -> {dur[k]*1000/max(1,inner[k]+leafph[k]):.0f} ns per round trip; all waves: {dur.sum()*1000/(inner.sum()+leafph.sum()):.0f} ns per round trip")
    print(f"     kernel span {span:8.1f} us  waves {len(t0)}  wave dur mean {dur.mean():7.1f} p50 {np.percentile(dur,50):6.1f} p99 {np.percentile(dur,99):7.1f} max {dur.max():7.1f} us | resident waves over time: {occ}")
    print(f"  {tag:10s} steps mean {steps.mean():7.1f} p99 {np.percentile(steps,99):5.0f} p99.9 {np.percentile(steps,99.9):5.0f} max {steps.max():6.0f} | leaves {leaves.mean():5.1f} tris {tris.mean():5.1f} | wave-max mean {wave_max.mean():7.1f}  lane util {util.mean()*100:5.1f}%  hit {(st[:,3]!=0xFFFFFFFF).mean()*100:.1f}%", flush=True)

if __name__ == "__main__":
    w, h = 1920, 1080
    sc = m.DragonScene((w, h))
    ctx = m.Context(0)
    rays = primary_rays(w, h, step=1)
    variants = [dict(builder=0), dict(builder=1), dict(builder=1, ploc_radius=8), dict(builder=1, ploc_radius=32), dict(builder=1, max_leaf=8), dict(builder=1, max_leaf=2), dict(builder=1, cost_trav=2.0)]
    if len(sys.argv) > 1:
        variants = [json.loads(a) for a in sys.argv[1:]]
    for opts in variants:
        ds = m.DeviceScene(ctx, sc, opts)
        s = ds.stats
        print(f"{opts}: nodes={s.bvh_nodes} leaves={s.bvh_leaves} depth={s.max_depth} sah={s.sah_cost:.2f} build={s.build_ms:.1f}ms", flush=True)
        st = ds.traversal_stats(rays)
        report("primary", st)
        hit = ds.intersect_closest(rays)
        ok = hit["type"] == 1
        P = rays[ok, 0:3] + rays[ok, 4:7] * hit["distance"][ok, None]
        rng = np.random.default_rng(0)
        d = rng.normal(size=P.shape).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True); d[:, 1] = np.abs(d[:, 1])
        n = (len(P) // 64) * 64
        r2 = np.zeros((n, 8), np.float32); r2[:, 0:3] = P[:n] + np.array([0, 2e-3, 0], np.float32); r2[:, 4:7] = d[:n]; r2[:, 7] = np.inf
        report("diffuse", ds.traversal_stats(r2))
        L = np.array([0, 1.98, 0], np.float32) + rng.uniform(-0.25, 0.25, (n, 3)).astype(np.float32) * np.array([1, 0, 1], np.float32)
        dl = L - r2[:, 0:3]; dist = np.linalg.norm(dl, axis=1); dl /= dist[:, None]
        r3 = r2.copy(); r3[:, 4:7] = dl; r3[:, 7] = dist - 1e-3
        report("shadow", ds.traversal_stats(r3, any_hit=True))
        ds.close()
