// traverse_wide_bundle.h — the eight primary rays of ONE pixel (eight sub-frames of a pass) walk the 8-wide tree together, eight lanes per node.
// Another form of Apple's opaque `intersector.intersect` for the camera rays (Raytracing.metal:244 with the ray of :214-221); flattened scenes, passes of several frames.
//
// Why: with FrameParams::frame_bundle the lanes 8 b .. 8 b + 7 of a wave of k_shade<.., TRACE0 = 2> hold the rays of one pixel in eight consecutive sub-frames — one origin, directions
// that differ by the jitter inside the pixel.  They visit the same nodes, and the one-ray-per-lane walk (traverse_wide_lane) decodes and tests each of those nodes' eight boxes eight
// times over, ~250 VALU instructions a time.  Here a BUNDLE (those eight lanes) keeps one stack and one pending group: lane k decodes child k of the bundle's node and tests
// that ONE box against the interval of the bundle's reciprocal directions (common origin, every component of one sign: a lower bound of the entry distance and an upper bound of
// the exit distance over the eight rays), and the hit bits are OR-ed across the eight lanes with three DPP steps.  A child that ANY ray's own test would accept is accepted (and a
// few more: the walk only has to be conservative — the closest hit is the minimum over (t, id) whatever is visited); every ray then tests every triangle of the leaves its
// bundle enters, with its own direction and its own limit.  A node costs the wave ~90 instructions for eight bundles instead of ~250 for 64 rays in as many different nodes.
// A bundle whose directions change sign in a component (the image's centre row / column) falls back to one ray per lane from the root.
#pragma once
#include "traverse_wide.h"

namespace mrt {
namespace {
static_assert(!MRT_WIDE6, "traverse_wide_bundle.h reads the 80-byte node");

// reductions over the eight lanes of a bundle: xor 1, xor 2 (quad_perm), then lane i <-> 7 - i (row_half_mirror).  Called by EVERY lane of the wave.
template <int CTRL> MRT_DEV uint32_t b8_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
MRT_DEV float b8_min(float v) {
    v = fminf(v, __uint_as_float(b8_dpp<0xB1>(__float_as_uint(v)))); v = fminf(v, __uint_as_float(b8_dpp<0x4E>(__float_as_uint(v)))); return fminf(v, __uint_as_float(b8_dpp<0x141>(__float_as_uint(v))));
}
MRT_DEV float b8_max(float v) {
    v = fmaxf(v, __uint_as_float(b8_dpp<0xB1>(__float_as_uint(v)))); v = fmaxf(v, __uint_as_float(b8_dpp<0x4E>(__float_as_uint(v)))); return fmaxf(v, __uint_as_float(b8_dpp<0x141>(__float_as_uint(v))));
}
MRT_DEV uint32_t b8_or(uint32_t v) { v |= b8_dpp<0xB1>(v); v |= b8_dpp<0x4E>(v); return v | b8_dpp<0x141>(v); }

// EVERY lane of the wave calls this (lanes without a ray with active = false; they still test boxes for their bundle).  o must be the same for the eight lanes of a bundle.
// stack: the wave's LDS stack (depth x WIDE_STACK_LEVEL_BYTES); a bundle uses the column of its first lane.
template <bool SEED>
MRT_DEV bool traverse_wide_bundle(const SceneView &s, const bool active, const f3 o, const f3 d, const float tmax, const uint32_t seed_pk, TravHit &h, uint32_t *stack) {
    const uint32_t lane = threadIdx.x & 63, k = lane & 7u, col = lane & 56u;
    float best_t = active ? tmax : -1.0f;                        // (a lane without a ray can hit nothing)
    uint32_t best_pk = (SEED && active) ? seed_pk : 0xFFFFFFFFu;
    const float inf = __builtin_inff();
    const float dlo[3] = {b8_min(active ? d.x : inf), b8_min(active ? d.y : inf), b8_min(active ? d.z : inf)};
    const float dhi[3] = {b8_max(active ? d.x : -inf), b8_max(active ? d.y : -inf), b8_max(active ? d.z : -inf)};
    bool ok = b8_or(active ? 1u : 0u) != 0u && s.num_wnodes != 0;
    float ilo[3], ihi[3]; bool neg[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (!((dlo[a] > 1e-9f && dhi[a] > 1e-9f) || (dlo[a] < -1e-9f && dhi[a] < -1e-9f))) ok = false;      // a sign change (or a component at zero) inside the bundle
        neg[a] = dhi[a] < 0.0f;
        const float r0 = 1.0f / dhi[a], r1 = 1.0f / dlo[a];          // same sign: 1 / d falls as d grows
        ilo[a] = r0 - fabsf(r0) * 1e-5f; ihi[a] = r1 + fabsf(r1) * 1e-5f;
    }
    const uint32_t oct = (neg[0] ? 1u : 0u) | (neg[1] ? 2u : 0u) | (neg[2] ? 4u : 0u);
    const float oc[3] = {o.x, o.y, o.z};
    // the bundle's walk: one state, the same in its eight lanes (g_mask: imask | hit bits << 8 | stack depth << 16, as in traverse_wide_lane)
    uint32_t g_base = 0, g_mask = ok ? 0x100u : 0u, t_base = 0, t_mask = 0;
    bool done = !ok;
    for (;;) {
        if (__ballot(!done) == 0ull) break;
        const bool has_tri = !done && t_mask != 0u;
        const uint32_t t_rest = t_mask & (t_mask - 1u);
        bool want_node = !done && t_rest == 0u;
        uint32_t pending = 0;
        if (want_node) {
            if ((g_mask & 0xFF00u) == 0u) {
                const uint32_t sp = g_mask >> 16;
                if (sp == 0u) { want_node = false; if (!has_tri) done = true; }
                else { wstack_pop(stack, sp - 1u, col, g_base, g_mask); g_mask |= (sp - 1u) << 16; }
            }
            if (want_node) {
                const uint32_t hits = (g_mask >> 8) & 0xFFu;
                const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;       // nearest remaining child in (slot ^ octant) order
                g_mask &= ~(0x100u << b);
                const uint32_t slot = b ^ oct;
                pending = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));
            }
        }
        float4 r0, r1, r2, n0, n1;
        asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r2.x), "=v"(r2.y), "=v"(r2.z));
        asm volatile("" : "=v"(n0.x), "=v"(n0.y), "=v"(n0.z), "=v"(n0.w), "=v"(n1.x), "=v"(n1.y), "=v"(n1.z), "=v"(n1.w));
        r1.w = 0.0f; r2.w = 0.0f;
        uint32_t tri_pk = 0, q[6] = {0, 0, 0, 0, 0, 0};
        if (has_tri) {          // the bundle's next triangle: the same packet for its eight rays
            tri_pk = t_base + (uint32_t)__ffs((int)t_mask) - 1u; t_mask = t_rest;
            const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)tri_pk;
            r0 = pk[0]; r1 = pk[1]; r2 = pk[2];
        }
        if (want_node) {        // the bundle's next node: lane k takes child k's six plane bytes (f4 2 = qlo_x[8] qlo_y[8], f4 3 = qlo_z[8] qhi_x[8], f4 4 = qhi_y[8] qhi_z[8])
            const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending;
            n0 = nd[0]; n1 = nd[1];
            const uint8_t *__restrict__ pb = reinterpret_cast<const uint8_t *>(nd + 2) + k;
            q[0] = pb[0]; q[1] = pb[8]; q[2] = pb[16]; q[3] = pb[24];
            const uint8_t *__restrict__ pc = reinterpret_cast<const uint8_t *>(nd + WNODE_N4) + k;
            q[4] = pc[0]; q[5] = pc[8];
        }
        if (has_tri && active) {
            float t, U, V, ad;
            if (tri_test(r0, r1, r2, o, d, 0.0f, best_t, t, U, V, ad)) {
                bool better = t < best_t || best_pk == 0xFFFFFFFFu;
                if (!better) better = __float_as_uint(r0.w) < __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);      // t == best_t: ties go to the lowest id
                if (better) { best_t = t; best_pk = tri_pk; }
            }
        }
        const float tb = b8_max(active ? best_t : 0.0f);          // nothing beyond the farthest of the bundle's limits matters
        uint32_t nh_k = 0, th_k = 0;
        if (want_node) {
            const uint32_t ew = __float_as_uint(n0.w), imask = ew >> 24;
            const float sc[3] = {__builtin_ldexpf(1.0f, (int)(int8_t)(ew & 0xFFu)), __builtin_ldexpf(1.0f, (int)(int8_t)((ew >> 8) & 0xFFu)), __builtin_ldexpf(1.0f, (int)(int8_t)((ew >> 16) & 0xFFu))};
            const float org[3] = {n0.x, n0.y, n0.z};
            float enter = 0.0f, leave = tb;
            const bool boxed = q[0] <= q[3] && q[1] <= q[4] && q[2] <= q[5];          // (an empty slot has qlo = 255, qhi = 0)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float lo = __builtin_fmaf((float)q[a], sc[a], org[a]), hi = __builtin_fmaf((float)q[3 + a], sc[a], org[a]);
                const float pn = (neg[a] ? hi : lo) - oc[a], pf = (neg[a] ? lo : hi) - oc[a];
                // the least entry distance and the greatest exit distance over 1 / d in [ilo, ihi], padded by what the rays' own (rounded) tests may differ by
                const float en = pn * (pn >= 0.0f ? ilo[a] : ihi[a]), ex = pf * (pf >= 0.0f ? ihi[a] : ilo[a]);
                enter = fmaxf(enter, en - (fabsf(en) * 1e-4f + 1e-5f));
                leave = fminf(leave, ex + (fabsf(ex) * 1e-4f + 1e-5f));
            }
            if (boxed && enter <= leave) {
                const uint32_t meta = (k < 4u ? __float_as_uint(n1.z) : __float_as_uint(n1.w)) >> (8u * (k & 3u));
                nh_k = ((imask >> k) & 1u) << (k ^ oct);          // an internal child's meta byte is 0, a leaf child's imask bit is 0
                th_k = bfm_b32((meta >> 5) & 7u, meta);
            }
        }
        const uint32_t nh = b8_or(nh_k), th = b8_or(th_k);
        if (want_node) {
            uint32_t sp = g_mask >> 16;
            if ((g_mask & 0xFF00u) != 0u) { wstack_push(stack, sp, col, g_base, g_mask & 0xFFFFu); sp++; }     // siblings still to visit
            g_base = __float_as_uint(n1.x) & WNODE_BASE_MASK; g_mask = (sp << 16) | (nh << 8) | (__float_as_uint(n0.w) >> 24);
            t_base = __float_as_uint(n1.y); t_mask = th;
        }
        else if (!done && t_rest == 0u) done = true;        // no node left and this was the last pending triangle
    }
    bool fell_back = false, hit_fb = false;
    if (active && !ok) { fell_back = true; hit_fb = traverse_wide_lane<SEED>(s, o, d, tmax, seed_pk, h, stack); }      // (its stack columns are free: every bundle is done)
    if (fell_back) return hit_fb;
    h.t = best_t; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu; h.pk = best_pk;
    if (!active || best_pk == 0xFFFFFFFFu) return false;
    // id and barycentrics of the winning triangle: recomputed (same arithmetic) instead of living in four registers through the loop
    const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)best_pk;
    const float4 q0 = pk[0];
    float t_;
    (void)tri_test(q0, pk[1], pk[2], o, d, 0.0f, __builtin_inff(), t_, h.U, h.V, h.ad);
    h.gid = __float_as_uint(q0.w);
    return true;
}

}  // namespace
}  // namespace mrt
