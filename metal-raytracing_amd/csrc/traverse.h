// traverse.h — stackless rope traversal + Möller–Trumbore for gfx950 (replaces the two uses of Apple's
// opaque `intersector.intersect`, Raytracing.metal:244 closest / :367 any).
#pragma once
#include "scene_device.h"
#include "device_math.h"

namespace mrt {
namespace {

struct TravHit { float t, U, V, ad; uint32_t gid; };

MRT_DEV float safe_inv(float d) {
    float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    return 1.0f / a;
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

struct TravCounters { uint32_t steps, leaves, tris, wave_iters; };

constexpr int MAX_LEAF_BATCH = 4;     // triangles fetched per round trip at a leaf

// Stackless traversal of the rope layout (scene_device.h).  State per ray: the current node index
// and the best hit — no stack, no parent walk.  Closest hit = global min t, ties to the lowest gid,
// so the result does not depend on the visiting order.
// Loop shape ("while-while"): lanes walk inner nodes until each has a leaf whose box it hits (or is
// done); then the wave tests leaf triangles together, all of a leaf's packets fetched in one round
// trip.  Every step is one dependent memory round trip, so the chain length of the slowest ray in a
// wave — not arithmetic — sets the wave's time.
template <bool ANY, bool STATS = false>
MRT_DEV bool traverse(const SceneView &s, f3 o, f3 d, float tmin, float tmax, TravHit &h, TravCounters *tc = nullptr) {
    h.t = tmax; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu;
    if (s.num_nodes == 0) return false;
    const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);
    const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    const uint32_t *__restrict__ nodes_u = reinterpret_cast<const uint32_t *>(s.nodes);
    uint32_t cur = 0;
    for (;;) {
        uint32_t leaf_first = 0, leaf_count = 0;
        while (cur != NODE_TERM) {
            if (STATS) { tc->steps++; }
            const float4 n0 = s.nodes[4 * (size_t)cur + 0];
            const float4 n1 = s.nodes[4 * (size_t)cur + 1];
            const uint32_t esc = nodes_u[16 * (size_t)cur + 8 + oct];
            // conservative slab test: far side widened by ~4 ulp (Ize 2013), boxes padded at build time
            float tx0 = (n0.x - o.x) * ix, tx1 = (n1.x - o.x) * ix;
            float ty0 = (n0.y - o.y) * iy, ty1 = (n1.y - o.y) * iy;
            float tz0 = (n0.z - o.z) * iz, tz1 = (n1.z - o.z) * iz;
            float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
            float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1)) * 1.0000005f;
            tf = fminf(tf, h.t);
            const uint32_t a = __float_as_uint(n0.w), b = __float_as_uint(n1.w);
            const bool hit = tn <= tf;
            const bool leaf = (a & NODE_LEAF) != 0;
            const uint32_t child = ((b >> (24 + oct)) & 1u) ? (b & NODE_INDEX_MASK) : a;
            cur = (hit && !leaf) ? child : esc;
            if (hit && leaf) { leaf_first = a & 0x7FFFFFFFu; leaf_count = b; break; }
        }
        if (leaf_count == 0) break;
        if (STATS) { tc->leaves++; tc->tris += leaf_count; }
        for (uint32_t base = 0; base < leaf_count; base += MAX_LEAF_BATCH) {
            float4 pk[MAX_LEAF_BATCH][3];
#pragma unroll
            for (int k = 0; k < MAX_LEAF_BATCH; k++) {
                // clamp instead of predicate: the loads are issued unconditionally, back to back
                uint32_t idx = leaf_first + min(base + (uint32_t)k, leaf_count - 1);
                pk[k][0] = s.packets[3 * (size_t)idx + 0];
                pk[k][1] = s.packets[3 * (size_t)idx + 1];
                pk[k][2] = s.packets[3 * (size_t)idx + 2];
            }
#pragma unroll
            for (int k = 0; k < MAX_LEAF_BATCH; k++) {
                if (base + (uint32_t)k < leaf_count) {
                    float t, U, V, ad;
                    if (tri_test(pk[k][0], pk[k][1], pk[k][2], o, d, tmin, h.t, t, U, V, ad)) {
                        if (ANY) return true;
                        uint32_t gid = __float_as_uint(pk[k][0].w);
                        if (t < h.t || gid < h.gid) { h.t = t; h.U = U; h.V = V; h.ad = ad; h.gid = gid; }   // t <= h.t here
                    }
                }
            }
        }
    }
    return h.gid != 0xFFFFFFFFu;
}

}  // namespace
}  // namespace mrt
