"""An INDEPENDENT float64 restatement of the reference's kernel, written straight from
`Raytracing.metal:41-56, 60-73, 78-88, 94-128, 132-147, 202-403` in numpy — TEST INFRASTRUCTURE.

Why it exists: the CPU oracle (oracle/mrt_oracle.cpp) and the HIP kernels share the "mrt-math v1"
design choices (fused Möller–Trumbore form, polynomial sincos, tie-break rule), so their bit-exact
agreement shows that two restatements of one design agree.  This file shares none of those choices:

  * arithmetic is float64 throughout (the kernel's fp32 rounding is NOT reproduced);
  * ray/triangle intersection is NOT Möller–Trumbore: the ray is intersected with the triangle's plane
    (t = n.(v0 - o) / n.d) and the barycentrics come from the 2x2 Gram system of the edge vectors;
    every triangle of the scene is tested for every ray (no acceleration structure);
  * sin/cos are numpy's, Halton digits come from Python integer arithmetic.

It is compared with the oracle under a tolerance (tests/test_independent_f64.py).  Nothing here can pin
the oracle against the Metal renderer itself — no image of the reference exists and its seeds are random
(`Renderer.swift:259`); DESIGN.md §2 says so.
"""
import numpy as np

PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]


def halton(i, d):
    """Raytracing.metal:41-56 in exact-digit float64."""
    b = PRIMES[d]
    f, r = 1.0, 0.0
    i = int(i)
    while i > 0:
        f /= b
        r += f * (i % b)
        i //= b
    return r


class Triangles:
    """World-space triangles of a flattened scene (positions transformed by each mesh's 4x4, column-major
    [col][row] as `Mesh.transform`), plus what shading needs: per-vertex object-space normals, the instance's
    3x3, the submesh's base colour, and (instance, geometry, primitive) ids in the reference's order."""

    def __init__(self, meshes):
        v0, v1, v2, n0, n1, n2, m3, col, ids = [], [], [], [], [], [], [], [], []
        for inst, (pos, nrm, xf, subs) in enumerate(meshes):
            M = np.asarray(xf, np.float64).reshape(4, 4).T          # M[row][col]
            P = np.asarray(pos, np.float64) @ M[:3, :3].T + M[:3, 3]
            N = np.asarray(nrm, np.float64)
            for geom, (idx, mat) in enumerate(subs):
                idx = np.asarray(idx, np.int64).reshape(-1, 3)
                v0.append(P[idx[:, 0]]); v1.append(P[idx[:, 1]]); v2.append(P[idx[:, 2]])
                n0.append(N[idx[:, 0]]); n1.append(N[idx[:, 1]]); n2.append(N[idx[:, 2]])
                m3.append(np.broadcast_to(M[:3, :3], (len(idx), 3, 3)))
                col.append(np.broadcast_to(np.array([mat.baseColor.x, mat.baseColor.y, mat.baseColor.z], np.float64), (len(idx), 3)))
                ids.append(np.c_[np.full(len(idx), inst), np.full(len(idx), geom), np.arange(len(idx))])
        cat = lambda a, shape: np.concatenate(a) if a else np.zeros(shape)
        self.v0, self.v1, self.v2 = cat(v0, (0, 3)), cat(v1, (0, 3)), cat(v2, (0, 3))
        self.n0, self.n1, self.n2 = cat(n0, (0, 3)), cat(n1, (0, 3)), cat(n2, (0, 3))
        self.m3, self.color, self.ids = cat(m3, (0, 3, 3)), cat(col, (0, 3)), cat(ids, (0, 3)).astype(np.int64)
        self.e1, self.e2 = self.v1 - self.v0, self.v2 - self.v0
        self.n = np.cross(self.e1, self.e2)
        # Gram matrix of the edges (for the barycentrics of a point in the triangle's plane)
        self.g11 = (self.e1 * self.e1).sum(1); self.g12 = (self.e1 * self.e2).sum(1); self.g22 = (self.e2 * self.e2).sum(1)
        self.gdet = self.g11 * self.g22 - self.g12 * self.g12

    def __len__(self):
        return len(self.v0)

    def intersect(self, o, d, tmax, chunk=None):
        """All rays against all triangles.  Returns per ray (t, u, v, tri) of the nearest hit with 0 <= t <= tmax
        (tri = -1: miss), and the runner-up distance `t2` (inf if none) so callers can tell near-ties apart."""
        R = len(o)
        if chunk is None:
            chunk = int(max(8, min(2048, 8_000_000 // max(1, len(self)))))      # rays per block: a handful of (rays x triangles) float64 temporaries of <= 64 MB each
        best_t = np.full(R, np.inf); best_u = np.zeros(R); best_v = np.zeros(R); best_i = np.full(R, -1, np.int64); second = np.full(R, np.inf)
        ok_tri = self.gdet > 0                                     # degenerate triangles have no interior
        for a in range(0, R, chunk):
            oo, dd, tm = o[a:a + chunk, None, :], d[a:a + chunk, None, :], np.asarray(tmax)[a:a + chunk, None]
            den = (self.n[None] * dd).sum(-1)
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (self.n[None] * (self.v0[None] - oo)).sum(-1) / den
                p = oo + t[..., None] * dd - self.v0[None]
                b1, b2 = (p * self.e1[None]).sum(-1), (p * self.e2[None]).sum(-1)
                u = (self.g22 * b1 - self.g12 * b2) / self.gdet
                v = (self.g11 * b2 - self.g12 * b1) / self.gdet
            hit = ok_tri[None] & (den != 0) & (t >= 0) & (t <= tm) & (u >= 0) & (v >= 0) & (u + v <= 1)
            t = np.where(hit, t, np.inf)
            k = np.argmin(t, axis=1)
            rows = np.arange(t.shape[0])
            bt = t[rows, k]
            t[rows, k] = np.inf
            second[a:a + chunk] = t.min(axis=1)
            h = np.isfinite(bt)
            best_t[a:a + chunk] = bt
            best_u[a:a + chunk] = np.where(h, u[rows, k], 0.0); best_v[a:a + chunk] = np.where(h, v[rows, k], 0.0)
            best_i[a:a + chunk] = np.where(h, k, -1)
        return best_t, best_u, best_v, best_i, second


def _normalize(v):
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


def _f3(x):
    return np.array([x.x, x.y, x.z], np.float64)


def render_frame(tris, lights, camera, width, height, seeds, frame_index, pixels, max_bounces=3):
    """radiance (len(pixels), 3) of one frame for the pixels [(x, y), ...] — Raytracing.metal:202-392 in float64.
    `seeds[y, x]` is the random texture (Renderer.swift:246-274)."""
    px = np.asarray(pixels, np.int64)
    n = len(px)
    idx = np.array([int(seeds[y, x]) + int(frame_index) for x, y in px])
    H = lambda dim: np.array([halton(i, dim) for i in idx])
    pixel = px.astype(np.float64) + np.c_[H(0), H(1)]                                   # :202-204
    uv = pixel / np.array([width, height], np.float64) * 2.0 - 1.0                      # :207-208
    cam_r, cam_u, cam_f, cam_p = _f3(camera.right), _f3(camera.up), _f3(camera.forward), _f3(camera.position)
    d = _normalize(uv[:, :1] * cam_r + uv[:, 1:] * cam_u + cam_f)                        # :216-218
    o = np.broadcast_to(cam_p, (n, 3)).copy()
    color = np.ones((n, 3)); acc = np.zeros((n, 3))
    alive = np.ones(n, bool)
    margin = np.full(n, np.inf)          # how far each path stayed from a discrete decision flipping (hit/miss, which triangle, light pick)
    nl = len(lights)
    for bounce in range(max_bounces):                                                    # :237
        ia = np.nonzero(alive)[0]
        if len(ia) == 0:
            break
        t, bu, bv, ti, t2 = tris.intersect(o[ia], d[ia], np.full(len(ia), np.inf))       # :244
        hit = ti >= 0
        alive[ia[~hit]] = False                                                          # :246-247
        ia, t, bu, bv, ti, t2 = ia[hit], t[hit], bu[hit], bv[hit], ti[hit], t2[hit]
        if len(ia) == 0:
            break
        # distance from a decision boundary: a triangle edge, or a second surface at nearly the same distance
        margin[ia] = np.minimum(margin[ia], np.minimum(np.minimum(bu, bv), 1.0 - bu - bv))
        margin[ia] = np.minimum(margin[ia], (t2 - t) / np.maximum(t, 1e-9))
        P = o[ia] + d[ia] * t[:, None]                                                   # :261
        n_obj = bu[:, None] * tris.n1[ti] + bv[:, None] * tris.n2[ti] + (1.0 - bu - bv)[:, None] * tris.n0[ti]   # :60-73
        nw = _normalize(np.einsum("kij,kj->ki", tris.m3[ti], n_obj))                     # :266-268
        surf = tris.color[ti]                                                            # :269
        ls = H(2 + bounce * 5 + 0)[ia]                                                   # :272
        li = np.minimum((ls * nl).astype(np.int64), nl - 1)                              # :273
        margin[ia] = np.minimum(margin[ia], np.abs(ls * nl - np.round(ls * nl)) if nl > 1 else np.inf)
        r1, r2 = H(2 + bounce * 5 + 1)[ia], H(2 + bounce * 5 + 2)[ia]
        L = np.zeros((len(ia), 3)); Lc = np.zeros((len(ia), 3)); dist = np.zeros(len(ia))
        for k, light in enumerate(lights):
            m = li == k
            if not m.any():
                continue
            lp, lcol = _f3(light.position), _f3(light.color)
            if light.type == 4:                                                          # area, :94-128
                ux, uy = r1[m] * 2.0 - 1.0, r2[m] * 2.0 - 1.0
                sp = lp + _f3(light.right) * ux[:, None] + _f3(light.up) * uy[:, None]
                l = sp - P[m]; dd = np.sqrt((l * l).sum(1)); inv = 1.0 / np.maximum(dd, 1e-3)
                l = l * inv[:, None]
                c = lcol * (inv * inv)[:, None] * np.clip((-l * _f3(light.forward)).sum(1), 0.0, 1.0)[:, None]
            elif light.type in (2, 3):                                                   # spot :292-316, point :317-322
                l = lp - P[m]; dd = np.sqrt((l * l).sum(1)); inv = 1.0 / np.maximum(dd, 1e-3)
                l = l * inv[:, None]
                c = lcol * (inv * inv)[:, None]
                if light.type == 2:
                    spot = (-l * _normalize(_f3(light.direction))).sum(1)
                    c = np.where((spot > np.cos(np.float64(light.coneAngle)))[:, None], c, 0.0)
                    margin[ia[m]] = np.minimum(margin[ia[m]], np.abs(spot - np.cos(np.float64(light.coneAngle))))
            else:                                                                        # sun :323-327
                l = np.broadcast_to(-_normalize(_f3(light.direction)), (m.sum(), 3)); dd = np.full(m.sum(), np.inf); c = np.broadcast_to(lcol, (m.sum(), 3))
            L[m], Lc[m], dist[m] = l, c, dd
        Lc = Lc * np.clip((nw * L).sum(1), 0.0, 1.0)[:, None] * nl                       # :331, :335
        color[ia] = color[ia] * surf                                                     # :339
        want = np.sqrt((Lc * Lc).sum(1)) > 1e-4                                          # :341
        so = P + nw * 1e-3                                                               # :350
        if want.any():
            st, _, _, si, _ = tris.intersect(so[want], L[want], dist[want] - 1e-3)       # :356-367
            vis = si < 0
            w = np.nonzero(want)[0]
            acc[ia[w[vis]]] += (Lc[w[vis]] * color[ia[w[vis]]])                          # :372
        h3, h4 = H(2 + bounce * 5 + 3)[ia], H(2 + bounce * 5 + 4)[ia]                    # :384-385
        phi = 2.0 * np.pi * h3
        ct = np.sqrt(h4); st_ = np.sqrt(1.0 - ct * ct)
        s = np.c_[st_ * np.cos(phi), ct, st_ * np.sin(phi)]                              # :78-88
        right = _normalize(np.cross(nw, np.array([0.0072, 1.0, 0.0034])))                # :132-147
        fwd = np.cross(right, nw)
        d[ia] = s[:, :1] * right + s[:, 1:2] * nw + s[:, 2:] * fwd
        o[ia] = so                                                                       # :390
    return acc, margin
