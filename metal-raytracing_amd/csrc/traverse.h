// traverse.h — stackless rope traversal + Möller–Trumbore for gfx950 (replaces the two uses of Apple's
// opaque `intersector.intersect`, Raytracing.metal:244 closest / :367 any).
#pragma once
#include "scene_device.h"
#include "device_math.h"

namespace mrt {
namespace {

struct TravHit { float t, U, V, ad; uint32_t gid; uint32_t pk = 0xFFFFFFFFu; };      // pk: packet of the hit (rope traversal with SEED only)

MRT_DEV float safe_inv(float d) {
    float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    return 1.0f / a;
}

// 1 / d for the slab tests of the stream and the two-level traversal: v_rcp_f32 (1 ulp) + one Newton step instead of the IEEE division sequence (13 instead of
// 37 issue cycles, three per refilled ray).  Only box tests see it — they have to be conservative, not exact: the error (< 1 ulp) is two
// orders of magnitude below the build-time padding of the leaf boxes (1e-5 |coord| + 1e-6) and the far side is widened by 4 ulp.  The
// triangle test, hence the image, does not depend on it.
MRT_DEV float box_inv(float d) {
    const float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    const float r = __builtin_amdgcn_rcpf(a);
    return __builtin_fmaf(r, __builtin_fmaf(-a, r, 1.0f), r);
}

// One triangle, Möller–Trumbore in the fused mrt-math form; division only after the barycentric
// tests pass.  Returns true when 0 <= tmin <= t <= lim.
MRT_DEV bool tri_test(float4 p0, float4 p1, float4 p2, f3 o, f3 d, float tmin, float lim, float &t, float &U, float &V, float &ad) {
    f3 v0 = mk3(p0), e1 = mk3(p1), e2 = mk3(p2);
    f3 pv = fcross(d, e2);
    float det = fdot(e1, pv);
    if (!(det != 0.0f)) return false;
    ad = fabsf(det);
    uint32_t sgn = __float_as_uint(det) & 0x80000000u;
    f3 tv = o - v0;
    U = xorsign(fdot(tv, pv), sgn);
    if (!(U >= 0.0f && U <= ad)) return false;
    f3 q = fcross(tv, e1);
    V = xorsign(fdot(d, q), sgn);
    if (!(V >= 0.0f && U + V <= ad)) return false;
    float T = xorsign(fdot(e2, q), sgn);
    t = T / ad;
    return t >= tmin && t <= lim;
}

struct TravCounters { uint32_t steps, leaves, tris, wave_iters; int alu_dup = 0, mem_dup = 0; float sink = 0.0f;
                      uint32_t empty = 0, stale = 0; };     // 8-wide layout: node visits in which no child was hit / whose own grid box already lay beyond the best hit

// Stackless traversal of the rope layout (scene_device.h).  State per ray: the next node, the pending
// triangle range of the current leaf and the best hit — no stack, no parent walk.  Closest hit =
// global min t, ties to the lowest gid, so the result does not depend on the visiting order.
//
// Loop shape: ONE kind of iteration.  Every iteration each live lane issues the same three 16-byte
// loads from its own base — a node (box lo|a, box hi|b, the escape quad of its octant) or one triangle
// packet (v0|gid, e1, e2) — so a wave iteration is exactly one memory round trip in which EVERY lane
// advances (a box test or a triangle test).  Measured on MI355X the wave time is (iterations of its
// slowest lane) x (round-trip latency); a split inner-node / leaf loop made the slowest wave iterate
// 4x more often than any of its lanes needed.
// SEED: `h` already holds a candidate hit (or t = tmax, gid = none) that the walk has to beat — a correct guess makes every box behind it a miss
// from the first step on; the result is the same minimum over (t, id) either way.  h.pk then tracks the packet of the winner.
template <bool ANY, bool STATS = false, bool RUNTIME_ANY = false, bool SEED = false>
MRT_DEV bool traverse(const SceneView &s, f3 o, f3 d, float tmin, float tmax, TravHit &h, TravCounters *tc = nullptr, bool any_rt = false) {
    if (!SEED) { h.t = tmax; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu; }
    if (s.num_nodes == 0) return false;
    const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);
    // slab planes as one fma each: t = plane * inv - o * inv.  The extra rounding (<= 1 ulp of o*inv, i.e.
    // <= 6e-8 * |o| along the axis) is far inside the build-time padding of the boxes (1e-6 + 1e-5 * |coord|).
    const float nox = -(o.x * ix), noy = -(o.y * iy), noz = -(o.z * iz);
    const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    const uint32_t esc_off = 32u + ((oct >> 2) << 4), esc_lane = oct & 3u;
    // nodes and packets live in one allocation: one wave-uniform base + a 32-bit byte offset per lane
    const char *__restrict__ base = reinterpret_cast<const char *>(s.nodes);
    const uint32_t pk0 = (uint32_t)(reinterpret_cast<const char *>(s.packets) - base);
    uint32_t cur = 0;                 // next node, NODE_TERM when the walk is over
    uint32_t tri = 0, tri_end = 0;    // pending packets of the current leaf
    for (;;) {
        const bool do_tri = tri < tri_end;
        if (!do_tri && cur == NODE_TERM) break;
        if (STATS) { if (do_tri) tc->tris++; else tc->steps++; if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) tc->wave_iters++; }
        const uint32_t off = do_tri ? pk0 + tri * 48u : cur << 6;
        const float4 r0 = *reinterpret_cast<const float4 *>(base + off);
        const float4 r1 = *reinterpret_cast<const float4 *>(base + off + 16u);
        const float4 r2 = *reinterpret_cast<const float4 *>(base + off + (do_tri ? 32u : esc_off));
        if (do_tri) {
            tri++;
            float t, U, V, ad;
            if (tri_test(r0, r1, r2, o, d, tmin, h.t, t, U, V, ad)) {
                if (ANY || (RUNTIME_ANY && any_rt)) return true;
                const uint32_t gid = __float_as_uint(r0.w);
                if (t < h.t || gid < h.gid) { h.t = t; h.U = U; h.V = V; h.ad = ad; h.gid = gid; if (SEED) h.pk = tri - 1u; }   // t <= h.t here
            }
        } else {
            // conservative slab test: far side widened by ~4 ulp (Ize 2013), boxes padded at build time
            float tx0 = __builtin_fmaf(r0.x, ix, nox), tx1 = __builtin_fmaf(r1.x, ix, nox);
            float ty0 = __builtin_fmaf(r0.y, iy, noy), ty1 = __builtin_fmaf(r1.y, iy, noy);
            float tz0 = __builtin_fmaf(r0.z, iz, noz), tz1 = __builtin_fmaf(r1.z, iz, noz);
            float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
            float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1)) * 1.0000005f;
            tf = fminf(tf, h.t);
            if (STATS) {      // bottleneck probes (diagnostics only): repeat the box arithmetic / the node fetch
                for (int r = 0; r < tc->alu_dup; r++) {
                    float e = 1e-9f * (float)(r + 1);
                    float ux0 = (r0.x - o.x + e) * ix, ux1 = (r1.x - o.x - e) * ix, uy0 = (r0.y - o.y + e) * iy, uy1 = (r1.y - o.y - e) * iy, uz0 = (r0.z - o.z + e) * iz, uz1 = (r1.z - o.z - e) * iz;
                    float un = fmaxf(fmaxf(fminf(ux0, ux1), fminf(uy0, uy1)), fmaxf(fminf(uz0, uz1), tmin));
                    float uf = fminf(fminf(fmaxf(ux0, ux1), fmaxf(uy0, uy1)), fmaxf(uz0, uz1)) * 1.0000005f;
                    tc->sink += (un <= fminf(uf, h.t)) ? 1.0f : 0.0f;
                }
                for (int r = 0; r < tc->mem_dup; r++) {
                    const float4 *__restrict__ rr = s.nodes + 4 * (size_t)((cur + 7919u * (r + 1)) % s.num_nodes);
                    tc->sink += rr[0].x + rr[1].y + rr[2].z;
                }
            }
            const uint32_t a = __float_as_uint(r0.w), b = __float_as_uint(r1.w);
            const float escf = esc_lane == 0 ? r2.x : esc_lane == 1 ? r2.y : esc_lane == 2 ? r2.z : r2.w;
            const uint32_t esc = __float_as_uint(escf);
            const bool hit = tn <= tf;
            const bool leaf = (a & NODE_LEAF) != 0;
            const uint32_t child = ((b >> (24 + oct)) & 1u) ? (b & NODE_INDEX_MASK) : a;
            cur = (hit && !leaf) ? child : esc;
            if (hit && leaf) { if (STATS) tc->leaves++; tri = a & 0x7FFFFFFFu; tri_end = tri + b; }
        }
    }
    return h.gid != 0xFFFFFFFFu;
}

}  // namespace
}  // namespace mrt
