// traverse_wide_tile.h — the primary rays of an 8x8 tile walk the TOP of the 8-wide tree together.
// Another form of Apple's opaque `intersector.intersect` for the camera rays (Raytracing.metal:244 with the ray of :214-221); flattened scenes.
//
// Why: the 64 primary rays of a wave leave one point in nearly one direction, and 87 % of their node visits are in the top four levels of the tree
// (profiles/r05_lds_top_ab.txt) — the same few nodes, tested 64 times over, ~250 VALU instructions a time.  Here the wave tests those nodes ONCE, against the
// bundle: lane (n, k) takes child k of node n (eight nodes per step), decodes that one box and tests it against the interval of the tile's reciprocal directions
// (common origin, every component of one sign) — a lower bound of the entry distance and an upper bound of the exit distance over all 64 rays, so a child that ANY
// ray's own test would accept is accepted (and a few more: the walk only has to be conservative, the closest hit is the minimum over (t, id) whatever is visited).
// Leaf children found on the way (walls, floor: the large triangles live up there) are tested by every ray of the tile; what is left after TILE_LEVELS levels is the
// tile's FRONT — the subtrees it may enter, nearest first — from which every ray continues on its own (traverse_wide_lane<.., ROOTS>), skipping those beyond its hit.
// A tile whose directions change sign in a component, an overfull list or a tree too shallow to borrow the stack as scratch: the rays walk from the root as before.
#pragma once
#include "traverse_wide.h"

namespace mrt {
namespace {

#ifndef MRT_TILE_LEVELS
#define MRT_TILE_LEVELS 4          // levels of the tree walked by the bundle (their children form the front)
#endif
constexpr uint32_t TILE_FRONT_WORDS = 128;      // per wave, in LDS: 64 {node, entry distance} pairs

MRT_DEV float tile_wave_min(float v) { for (int o_ = 32; o_ > 0; o_ >>= 1) v = fminf(v, __shfl_xor(v, o_)); return v; }
MRT_DEV float tile_wave_max(float v) { for (int o_ = 32; o_ > 0; o_ >>= 1) v = fmaxf(v, __shfl_xor(v, o_)); return v; }

// EVERY lane of the wave calls this (inactive pixels with active = false).  o must be the same for all lanes (the camera position).
// stack: the wave's LDS stack (stack_words words, used as scratch for the node lists while the bundle walks); front: TILE_FRONT_WORDS words of LDS of this wave.
template <bool SEED>
MRT_DEV bool traverse_wide_tile(const SceneView &s, const bool active, const f3 o, const f3 d, const float tmax, const uint32_t seed_pk, TravHit &h, uint32_t *stack, const uint32_t stack_words, uint32_t *front) {
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    float best_t = active ? tmax : -1.0f;                        // (an inactive lane can hit nothing and reach no subtree)
    uint32_t best_pk = SEED ? seed_pk : 0xFFFFFFFFu;
    uint32_t nroots = 0;
    // the bundle: per component the interval of 1 / d over the tile's rays, widened outwards
    const float inf = __builtin_inff();
    const float dlo[3] = {tile_wave_min(active ? d.x : inf), tile_wave_min(active ? d.y : inf), tile_wave_min(active ? d.z : inf)};
    const float dhi[3] = {tile_wave_max(active ? d.x : -inf), tile_wave_max(active ? d.y : -inf), tile_wave_max(active ? d.z : -inf)};
    bool bundle = s.num_wnodes != 0 && stack_words >= 128u && __ballot(active) != 0ull;
    float ilo[3], ihi[3]; bool neg[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (!((dlo[a] > 1e-9f && dhi[a] > 1e-9f) || (dlo[a] < -1e-9f && dhi[a] < -1e-9f))) bundle = false;      // a sign change (or a component at zero) inside the tile
        neg[a] = dhi[a] < 0.0f;
        const float r0 = 1.0f / dhi[a], r1 = 1.0f / dlo[a];          // same sign: 1 / d falls as d grows
        ilo[a] = r0 - fabsf(r0) * 1e-5f; ihi[a] = r1 + fabsf(r1) * 1e-5f;
    }
    uint32_t tri_reg = 0, ntri = 0;                              // the leaf triangles found by the bundle: lane i holds packet i
    if (bundle) {
        const float tb = tile_wave_max(active ? best_t : 0.0f);  // nothing beyond the farthest of the rays' limits matters
        uint32_t *listA = stack, *listB = stack + 64;
        uint32_t nA = 1, nfr = 0;
        if (lane == 0) listA[0] = 0;
        bool overflow = false;
        for (int level = 0; level < MRT_TILE_LEVELS && nA != 0u && !overflow; level++) {
            const bool last = level + 1 == MRT_TILE_LEVELS;
            uint32_t nB = 0;
            for (uint32_t base = 0; base < nA && !overflow; base += 8u) {
                const uint32_t ni = base + (lane >> 3), k = lane & 7u;
                const bool valid = ni < nA;
                const uint32_t node = valid ? listA[ni] : 0u;
                const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)node;
                const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3], n4 = nd[WNODE_N4];
                const uint32_t ew = __float_as_uint(n0.w), imask = ew >> 24;
                // child k's six plane bytes: f4 2 = qlo_x[8] qlo_y[8], f4 3 = qlo_z[8] qhi_x[8], f4 4 = qhi_y[8] qhi_z[8]
                const uint32_t sh = 8u * (k & 3u); const bool hiw = k >= 4u;
                const uint32_t qlx = ((hiw ? __float_as_uint(n2.y) : __float_as_uint(n2.x)) >> sh) & 0xFFu, qly = ((hiw ? __float_as_uint(n2.w) : __float_as_uint(n2.z)) >> sh) & 0xFFu;
                const uint32_t qlz = ((hiw ? __float_as_uint(n3.y) : __float_as_uint(n3.x)) >> sh) & 0xFFu, qhx = ((hiw ? __float_as_uint(n3.w) : __float_as_uint(n3.z)) >> sh) & 0xFFu;
                const uint32_t qhy = ((hiw ? __float_as_uint(n4.y) : __float_as_uint(n4.x)) >> sh) & 0xFFu, qhz = ((hiw ? __float_as_uint(n4.w) : __float_as_uint(n4.z)) >> sh) & 0xFFu;
                const float sx = __builtin_ldexpf(1.0f, (int)(int8_t)(ew & 0xFFu)), sy = __builtin_ldexpf(1.0f, (int)(int8_t)((ew >> 8) & 0xFFu)), sz = __builtin_ldexpf(1.0f, (int)(int8_t)((ew >> 16) & 0xFFu));
                const float lo[3] = {__builtin_fmaf((float)qlx, sx, n0.x), __builtin_fmaf((float)qly, sy, n0.y), __builtin_fmaf((float)qlz, sz, n0.z)};
                const float hi[3] = {__builtin_fmaf((float)qhx, sx, n0.x), __builtin_fmaf((float)qhy, sy, n0.y), __builtin_fmaf((float)qhz, sz, n0.z)};
                const float oc[3] = {o.x, o.y, o.z};
                float enter = 0.0f, leave = tb;
                bool boxed = qlx <= qhx && qly <= qhy && qlz <= qhz;          // (an empty slot has qlo = 255, qhi = 0)
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float pn = (neg[a] ? hi[a] : lo[a]) - oc[a], pf = (neg[a] ? lo[a] : hi[a]) - oc[a];
                    // the least entry distance and the greatest exit distance over 1 / d in [ilo, ihi], padded by what the rays' own (rounded) tests may differ by
                    const float en = pn * (pn >= 0.0f ? ilo[a] : ihi[a]), ex = pf * (pf >= 0.0f ? ihi[a] : ilo[a]);
                    enter = fmaxf(enter, en - (fabsf(en) * 1e-4f + 1e-5f));
                    leave = fminf(leave, ex + (fabsf(ex) * 1e-4f + 1e-5f));
                }
                const bool hit = valid && boxed && enter <= leave;
                const bool inner = ((imask >> k) & 1u) != 0u;
                const uint32_t meta = (k < 4u ? __float_as_uint(n1.z) : __float_as_uint(n1.w)) >> sh;
                const uint32_t tcnt = inner ? 0u : (meta >> 5) & 7u, tfirst = __float_as_uint(n1.y) + (meta & 31u);
                // inner children: the next level's list, or the front
                const unsigned long long m_in = __ballot(hit && inner);
                const uint32_t c_in = (uint32_t)__popcll(m_in), r_in = (uint32_t)__popcll(m_in & lt);
                const uint32_t child = (__float_as_uint(n1.x) & WNODE_BASE_MASK) + (uint32_t)__popc(imask & ((1u << k) - 1u));
                if (!last) {
                    if (nB + c_in > 64u) overflow = true;
                    else { if (hit && inner) listB[nB + r_in] = child; nB += c_in; }
                } else {
                    if (nfr + c_in > 64u) overflow = true;
                    else { if (hit && inner) { front[2u * (nfr + r_in)] = child; front[2u * (nfr + r_in) + 1u] = __float_as_uint(enter); } nfr += c_in; }
                }
                // leaf children: their triangles join the tile's list (few: a short scalar loop)
                for (unsigned long long m = __ballot(hit && tcnt != 0u); m != 0ull && !overflow; m &= m - 1ull) {
                    const int l = __ffsll((long long)m) - 1;
                    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)tcnt, l), f = (uint32_t)__builtin_amdgcn_readlane((int)tfirst, l);
                    if (ntri + c > 64u) overflow = true;
                    else { if (lane >= ntri && lane < ntri + c) tri_reg = f + (lane - ntri); ntri += c; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t *t_ = listA; listA = listB; listB = t_; nA = nB;
        }
        if (overflow) bundle = false;
        else {
            // the tile's large triangles, every ray against each (the packet's address is wave-uniform)
            for (uint32_t i = 0; i < ntri; i++) {
                const uint32_t pk_i = (uint32_t)__builtin_amdgcn_readlane((int)tri_reg, (int)i);
                const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)pk_i;
                const float4 r0 = pk[0], r1 = pk[1], r2 = pk[2];
                float t, U, V, ad;
                if (active && tri_test(r0, r1, r2, o, d, 0.0f, best_t, t, U, V, ad)) {
                    bool better = t < best_t || best_pk == 0xFFFFFFFFu;
                    if (!better) better = __float_as_uint(r0.w) < __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);      // t == best_t: ties go to the lowest id
                    if (better) { best_t = t; best_pk = pk_i; }
                }
            }
            // the front, nearest first, without what lies beyond every ray's hit: a rank sort over its (at most 64) entries
            __builtin_amdgcn_wave_barrier();
            const float tb2 = tile_wave_max(active ? best_t : 0.0f);
            const uint32_t e_node = lane < nfr ? front[2u * lane] : 0u;
            const float e_key = lane < nfr ? __uint_as_float(front[2u * lane + 1u]) : inf;
            const bool keep = lane < nfr && e_key <= tb2;
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nfr; j++) {
                const float kj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(e_key), (int)j));
                const bool keepj = __builtin_amdgcn_readlane((int)keep, (int)j) != 0;
                if (keepj && (kj < e_key || (kj == e_key && j < lane))) rank++;
            }
            __builtin_amdgcn_wave_barrier();
            if (keep) { front[2u * rank] = e_node; front[2u * rank + 1u] = __float_as_uint(e_key); }
            nroots = (uint32_t)__popcll(__ballot(keep));
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!bundle) {          // every ray from the root, as traverse_wide_lane alone would
        if (lane == 0) { front[0] = 0u; front[1] = 0u; }
        nroots = s.num_wnodes != 0 ? 1u : 0u;
        best_t = active ? tmax : -1.0f; best_pk = SEED ? seed_pk : 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
    }
    return traverse_wide_lane<true, true>(s, o, d, best_t, best_pk, h, stack, front, nroots);
}

}  // namespace
}  // namespace mrt
