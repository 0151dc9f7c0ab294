// traverse_wide.h — traversal of the 8-wide compressed layout (scene_device.h "wide node") with a short
// per-lane stack held in LDS.  Second backend for the two uses of Apple's opaque
// `intersector.intersect` (Raytracing.metal:244 closest, :367 any).
//
// Why it exists (measured on MI355X): with the binary rope layout every box test costs one
// dependent memory round trip of 3 x 16 B per lane, and the traversal kernels were bound by the vector
// memory path (TA/L1 request rate on divergent 16-B gathers) and by the length of the dependent chain of
// the slowest ray — not by arithmetic.  One 80-byte wide node tests eight quantised child boxes per
// round trip: ~4x fewer 16-B requests per box and ~3x fewer round trips per ray.  The stack holds
// (child_base, remaining hit bits, imask) entries, 5 B each, at most one entry per tree level; it lives in LDS
// ([level][lane], conflict free), never in scratch or HBM.
//
// The result is identical to the rope backend bit for bit: the closest hit is the global minimum t
// with ties to the lowest triangle id, and the quantised boxes only ever grow.
#pragma once
#include "traverse.h"
#include "traverse_instanced.h"

namespace mrt {
namespace {

// Compile-time switches kept for A/B builds (tools/build_variant.sh); the defaults are the measured best.
// (measured and removed, DESIGN.md §6: packed-FMA plane evaluation, two triangles per iteration, a lane quorum for the node branch, an inner / leaf branch per hit child)
#ifndef MRT_WIDE_SCALED
#define MRT_WIDE_SCALED 1   // stream kernels: box distances in units of the ray's limit, interval ends from the clamp modifier (wide_node_test<true>)
#endif
#ifndef MRT_WIDE_SPEC
#define MRT_WIDE_SPEC 1     // flattened scenes: a lane that still has triangles to test visits its next node anyway and keeps that node's triangles in a second group (below)
#endif
#ifndef MRT_WIDE_SPEC_TWO_LEVEL
#define MRT_WIDE_SPEC_TWO_LEVEL 0     // the same inside the BLASes of two-level scenes: measured twice, no gain
#endif
#ifndef MRT_COOP_MODE
#define MRT_COOP_MODE 3   // drain phase: idle lanes test the pending triangles of a straggler ray; bit 0 = any-hit owners, bit 1 = closest-hit owners (0 = off: A/B)
#endif
// -DMRT_DEBUG_BOUNDS (tools/build_variant.sh bounds "-DMRT_DEBUG_BOUNDS"): every node, packet and instance index the stream traversal is about
// to follow is checked against its array; the first violation is recorded (kind << 28 | index) and the index replaced by 0, and the host
// turns a non-zero record into MRT_ERR_STATE at the next wait (renderer.hip).  The release build has no such checks: the commit-time
// validator (two_level.hip validate_layout) is what keeps bad indices out of the arrays.
#ifdef MRT_DEBUG_BOUNDS
__device__ uint32_t g_bounds_violation = 0;
#define MRT_BOUND(idx, limit, kind) do { if ((idx) >= (limit)) { atomicMax(&g_bounds_violation, ((uint32_t)(kind) << 28) | min((uint32_t)(idx), 0x0FFFFFFFu)); (idx) = 0; } } while (0)
#else
#define MRT_BOUND(idx, limit, kind) do { } while (0)
#endif
// ((1 << width) - 1) << offset in one instruction (width, offset taken mod 32)
MRT_DEV uint32_t bfm_b32(uint32_t width, uint32_t offset) { uint32_t r; asm("v_bfm_b32 %0, %1, %2" : "=v"(r) : "v"(width), "v"(offset)); return r; }
MRT_DEV float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }   // -> v_cvt_f32_ubyteK

// LDS stack of one wave: per tree level 64 words {child_base << 8 | remaining hit bits} followed by 64 bytes {imask}
// = WIDE_STACK_LEVEL_BYTES (320) per level; 5 B per lane and level instead of 8 buys occupancy (LDS is what limits
// the waves per CU of the wide kernels: +2 KB per wave costs 5 % of the frame rate).
MRT_DEV void wstack_push(uint32_t *stack, uint32_t sp, uint32_t lane, uint32_t g_base, uint32_t g_mask) {
    stack[sp * (WIDE_STACK_LEVEL_BYTES / 4u) + lane] = (g_base << 8) | (g_mask >> 8);
    reinterpret_cast<uint8_t *>(stack)[sp * WIDE_STACK_LEVEL_BYTES + 256u + lane] = (uint8_t)g_mask;
}
MRT_DEV void wstack_pop(const uint32_t *stack, uint32_t sp, uint32_t lane, uint32_t &g_base, uint32_t &g_mask) {
    const uint32_t w = stack[sp * (WIDE_STACK_LEVEL_BYTES / 4u) + lane];
    const uint32_t im = reinterpret_cast<const uint8_t *>(stack)[sp * WIDE_STACK_LEVEL_BYTES + 256u + lane];
    g_base = w >> 8; g_mask = ((w & 0xFFu) << 8) | im;
}

// The eight quantised child boxes of one wide node against a ray: node_hits = bit (slot ^ octant) per internal child hit
// (so that ffs walks them front to back), tri_hits = bit k per packet tri_base + k of the leaf children hit.
// Plane distance t = q * (2^e * idir) + (p - o) * idir, one fma per plane; the decode error of the fused evaluation is far
// below the build-time padding of the leaf boxes, and the far side is widened by 4 ulp (Ize 2013).
//
// SCALED (the stream kernels; tmin = 0): the distances are computed in units of the ray's current limit — 1/d is multiplied by S = 1 / (tmax * (1 + 4 ulp)),
// limits clamped to [1e-6, 1e30] so that S stays finite for tmax = 0 and for tmax = inf — and the two ends of the ray's interval come from the
// free `clamp` output modifier ([0, 1]) instead of a v_max with 0 and a v_min with tmax per child (half-rate instructions):
//   near = max3(nx, ny, clamp(nz)) >= 0,   far = clamp(min3(fx, fy, fz) * (1 + 4 ulp)) <= 1,   hit <=> near < far.
// The comparison is strict because both ends clamp to the same value when the box lies behind the origin (far = 0 = near) or beyond the
// limit (near >= 1 = far); for a box that the exact test accepts the widened far side is strictly beyond the near side.  Per node this costs
// v_med3 + v_mul + v_rcp + 3 v_mul and saves 8 x (v_max + v_min).
// MRT_ROOT_AT_FETCH (the stream kernels with LDS extras): every ray's walk begins with the SAME node.  The batch prefetch — 64 rays, one per lane, every lane busy — tests the root's
// eight boxes for its rays there (the root's words are wave-uniform loads) and leaves the hit bits in LDS; a lane that takes a ray starts with the root's children as its group
// instead of spending its first iteration — a round trip, and a node test at the loop's ~55 % of the lanes — on node 0.
#ifndef MRT_ROOT_AT_FETCH
#define MRT_ROOT_AT_FETCH 1
#endif
template <bool SCALED = false>
MRT_DEV void wide_node_test(const float4 n0, const float4 n1, const float4 n2, const float4 n3, const float4 n4, const f3 o,
                            float ix, float iy, float iz, const bool nx, const bool ny, const bool nz, const uint32_t oct,
                            const float tmin, const float tmax, uint32_t &node_hits, uint32_t &tri_hits) {
    const uint32_t ew = __float_as_uint(n0.w);
    const uint32_t imask = ew >> 24;
    if (SCALED) {
        const float S = __builtin_amdgcn_rcpf(__builtin_amdgcn_fmed3f(tmax, 1e-6f, 1e30f) * 1.0000005f);
        ix *= S; iy *= S; iz *= S;
    }
    // 2^e * idir: sign-extended exponent byte (v_bfe_i32) + v_ldexp_f32 — two instructions per axis instead of shift, mask, multiply
    const float ax = __builtin_ldexpf(ix, (int)(int8_t)(ew & 0xFFu)), ay = __builtin_ldexpf(iy, (int)(int8_t)((ew >> 8) & 0xFFu)), az = __builtin_ldexpf(iz, (int)(int8_t)((ew >> 16) & 0xFFu));
    const float bx = (n0.x - o.x) * ix, by = (n0.y - o.y) * iy, bz = (n0.z - o.z) * iz;
    // near / far plane bytes per axis: swap lo and hi where the direction is negative
    const uint32_t lx0 = __float_as_uint(n2.x), lx1 = __float_as_uint(n2.y), ly0 = __float_as_uint(n2.z), ly1 = __float_as_uint(n2.w);
    const uint32_t lz0 = __float_as_uint(n3.x), lz1 = __float_as_uint(n3.y), hx0 = __float_as_uint(n3.z), hx1 = __float_as_uint(n3.w);
    const uint32_t hy0 = __float_as_uint(n4.x), hy1 = __float_as_uint(n4.y), hz0 = __float_as_uint(n4.z), hz1 = __float_as_uint(n4.w);
    const uint32_t nrx[2] = {nx ? hx0 : lx0, nx ? hx1 : lx1}, frx[2] = {nx ? lx0 : hx0, nx ? lx1 : hx1};
    const uint32_t nry[2] = {ny ? hy0 : ly0, ny ? hy1 : ly1}, fry[2] = {ny ? ly0 : hy0, ny ? ly1 : hy1};
    const uint32_t nrz[2] = {nz ? hz0 : lz0, nz ? hz1 : lz1}, frz[2] = {nz ? lz0 : hz0, nz ? lz1 : hz1};
    const uint32_t meta[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)};
    uint32_t nh = 0, th = 0;     // locals, not the reference parameters: `if (..) a |= x; else b |= y;` on references becomes a pointer select -> scratch
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int w = i >> 2, k = i & 3;
        float tn, tf;
        if (SCALED) {
            tn = fmaxf(fmaxf(__builtin_fmaf(ubyte_f(nrx[w], k), ax, bx), __builtin_fmaf(ubyte_f(nry[w], k), ay, by)),
                       __builtin_amdgcn_fmed3f(__builtin_fmaf(ubyte_f(nrz[w], k), az, bz), 0.0f, 1.0f));
            tf = __builtin_amdgcn_fmed3f(fminf(fminf(__builtin_fmaf(ubyte_f(frx[w], k), ax, bx), __builtin_fmaf(ubyte_f(fry[w], k), ay, by)),
                                               __builtin_fmaf(ubyte_f(frz[w], k), az, bz)) * 1.0000005f, 0.0f, 1.0f);
        } else {
            tn = fmaxf(fmaxf(__builtin_fmaf(ubyte_f(nrx[w], k), ax, bx), __builtin_fmaf(ubyte_f(nry[w], k), ay, by)),
                       fmaxf(__builtin_fmaf(ubyte_f(nrz[w], k), az, bz), tmin));
            tf = fminf(fminf(fminf(__builtin_fmaf(ubyte_f(frx[w], k), ax, bx), __builtin_fmaf(ubyte_f(fry[w], k), ay, by)),
                             __builtin_fmaf(ubyte_f(frz[w], k), az, bz)) * 1.0000005f, tmax);
        }
        if (SCALED ? tn < tf : tn <= tf) {
            // no inner branch: an internal child's meta byte is 0 (empty triangle range), a leaf child's imask bit is 0
            nh |= ((imask >> i) & 1u) << ((uint32_t)i ^ oct);
            th |= bfm_b32((meta[w] >> (8 * k + 5)) & 7u, meta[w] >> (8 * k));       // v_bfm_b32 reads the low 5 bits of the offset operand
        }
    }
    node_hits = nh; tri_hits = th;
}

// stack: depth x WIDE_STACK_LEVEL_BYTES of LDS for the wave (depth = the scene's wide-tree depth, <= WIDE_STACK)
template <bool ANY, bool STATS = false, bool RUNTIME_ANY = false>
MRT_DEV bool traverse_wide(const SceneView &s, f3 o, f3 d, float tmin, float tmax, TravHit &h, uint32_t *stack /* depth x WIDE_STACK_LEVEL_BYTES in LDS */, TravCounters *tc = nullptr, bool any_rt = false, uint32_t root = 0 /* the node the walk starts at: a BLAS root of a two-level scene, with the ray in that instance's object space */) {
    h.t = tmax; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu;
    if (s.num_wnodes == 0) return false;
    const uint32_t lane = threadIdx.x & 63;
    const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);
    const bool nx = d.x < 0.0f, ny = d.y < 0.0f, nz = d.z < 0.0f;
    const uint32_t oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
    uint32_t sp = 0;
    uint32_t g_base = 0, g_mask = 0;      // node group: (permuted hit bits << 8) | imask
    uint32_t t_base = 0, t_mask = 0;      // triangle group: bit k = packet t_base + k still to test
    uint32_t pending = root; bool have_pending = true;     // start by entering the root
    for (;;) {
        const bool do_tri = t_mask != 0;
        if (!do_tri && !have_pending) {
            if ((g_mask >> 8) == 0) {
                if (sp == 0) break;
                sp--;
                wstack_pop(stack, sp, lane, g_base, g_mask);
            }
            const uint32_t hits = g_mask >> 8;
            const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;       // nearest remaining child in (slot ^ octant) order
            g_mask &= ~(0x100u << b);
            const uint32_t slot = b ^ oct;
            pending = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));
            have_pending = true;
        }
        if (STATS) { if (do_tri) tc->tris++; else tc->steps++; if ((int)lane == __ffsll((long long)__ballot(1)) - 1) tc->wave_iters++; }
        if (do_tri) {
            const uint32_t k = (uint32_t)__ffs((int)t_mask) - 1u;
            t_mask &= t_mask - 1u;
            const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)(t_base + k);
            const float4 r0 = pk[0], r1 = pk[1], r2 = pk[2];
            float t, U, V, ad;
            if (tri_test(r0, r1, r2, o, d, tmin, h.t, t, U, V, ad)) {
                if (ANY || (RUNTIME_ANY && any_rt)) return true;
                const uint32_t gid = __float_as_uint(r0.w);
                if (t < h.t || gid < h.gid) { h.t = t; h.U = U; h.V = V; h.ad = ad; h.gid = gid; }
            }
        } else {
            const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending;
            const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3], n4 = nd[4];
            have_pending = false;
            uint32_t node_hits, tri_hits;
            wide_node_test(n0, n1, n2, n3, n4, o, ix, iy, iz, nx, ny, nz, oct, tmin, h.t, node_hits, tri_hits);
            if (STATS && tri_hits) tc->leaves++;
            if (STATS) {
                if ((node_hits | tri_hits) == 0u) tc->empty++;
                // the node's own box (its quantisation grid, p .. p + 255 * 2^e, which encloses every child) against the ray's current limit: a visit that a
                // distance kept with the stack entry would have skipped
                const uint32_t ew = __float_as_uint(n0.w);
                const float gx = __builtin_ldexpf(255.0f, (int)(int8_t)(ew & 0xFFu)), gy = __builtin_ldexpf(255.0f, (int)(int8_t)((ew >> 8) & 0xFFu)), gz = __builtin_ldexpf(255.0f, (int)(int8_t)((ew >> 16) & 0xFFu));
                const float x0 = (n0.x - o.x) * ix, x1 = (n0.x + gx - o.x) * ix, y0 = (n0.y - o.y) * iy, y1 = (n0.y + gy - o.y) * iy, z0 = (n0.z - o.z) * iz, z1 = (n0.z + gz - o.z) * iz;
                const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
                if (tn > h.t) tc->stale++;
            }
            if ((g_mask >> 8) != 0) { wstack_push(stack, sp, lane, g_base, g_mask); sp++; }     // siblings still to visit
            g_base = __float_as_uint(n1.x); g_mask = (node_hits << 8) | (__float_as_uint(n0.w) >> 24);
            t_base = __float_as_uint(n1.y); t_mask = tri_hits;
        }
    }
    return h.gid != 0xFFFFFFFFu;
}

// One ray per lane on the 8-wide layout, closest hit, the stream loop's iteration (node and triangle fetched in one round trip, scaled box test) without
// refill: for COHERENT rays — the primary rays of an 8x8 tile, traced inside k_shade<.., TRACE0 = 2> — whose lanes stay in step by themselves.
// SEED: `seed_pk` is a packet the caller has already tested (distance in tmax); the walk starts with it as its closest hit.  h.pk = packet of the final hit.
#ifndef MRT_LANE_HIT_LDS
#define MRT_LANE_HIT_LDS 1      // 1: the lane walk keeps U, V, |det|, id of its closest hit in four words of LDS (`hitw`, [4][64] behind the caller's stack) instead of testing the winner again at the end
#endif
template <bool SEED>
MRT_DEV bool traverse_wide_lane(const SceneView &s, const f3 o, const f3 d, float tmax, uint32_t seed_pk, TravHit &h, uint32_t *stack /* depth x WIDE_STACK_LEVEL_BYTES of LDS, this wave's */, float *hitw = nullptr, const TravHit *seed_hit = nullptr) {
    const uint32_t lane = threadIdx.x & 63;
    const float ix = box_inv(d.x), iy = box_inv(d.y), iz = box_inv(d.z);
    const bool nx = d.x < 0.0f, ny = d.y < 0.0f, nz = d.z < 0.0f;
    const uint32_t oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
    float best_t = tmax; uint32_t best_pk = SEED ? seed_pk : 0xFFFFFFFFu;
    if (MRT_LANE_HIT_LDS && hitw) {          // the seed's own U, V, |det|, id (the caller tested it) are the closest hit so far
        if (SEED && seed_hit && seed_pk != 0xFFFFFFFFu) { hitw[lane] = seed_hit->U; hitw[64u + lane] = seed_hit->V; hitw[128u + lane] = seed_hit->ad; hitw[192u + lane] = __uint_as_float(seed_hit->gid); }
    }
    uint32_t g_base = 0, g_mask = s.num_wnodes != 0 ? 0x100u : 0u, t_base = 0, t_mask = 0;      // the root as the only hit child of a pseudo group; g_mask: imask | hit bits << 8 | stack depth << 16
    for (;;) {
        const bool has_tri = t_mask != 0;
        const uint32_t t_rest = t_mask & (t_mask - 1u);
        bool want_node = t_rest == 0u;
        uint32_t pending = 0;
        if (want_node) {
            if ((g_mask & 0xFF00u) == 0) {
                const uint32_t sp = g_mask >> 16;
                if (sp == 0) { want_node = false; if (!has_tri) break; }
                else { wstack_pop(stack, sp - 1u, lane, g_base, g_mask); g_mask |= (sp - 1u) << 16; }
            }
            if (want_node) {
                const uint32_t hits = (g_mask >> 8) & 0xFFu;
                const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;       // nearest remaining child in (slot ^ octant) order
                g_mask &= ~(0x100u << b);
                const uint32_t slot = b ^ oct;
                pending = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));
            }
        }
        float4 r0, r1, r2, n0, n1, n2, n3, n4;
        asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r2.x), "=v"(r2.y), "=v"(r2.z));
        asm volatile("" : "=v"(n0.x), "=v"(n0.y), "=v"(n0.z), "=v"(n0.w), "=v"(n1.x), "=v"(n1.y), "=v"(n1.z), "=v"(n1.w), "=v"(n2.x), "=v"(n2.y), "=v"(n2.z), "=v"(n2.w));
        asm volatile("" : "=v"(n3.x), "=v"(n3.y), "=v"(n3.z), "=v"(n3.w), "=v"(n4.x), "=v"(n4.y), "=v"(n4.z), "=v"(n4.w));
        r1.w = 0.0f; r2.w = 0.0f;
        uint32_t tri_pk = 0;
        if (has_tri) {
            tri_pk = t_base + (uint32_t)__ffs((int)t_mask) - 1u; t_mask = t_rest;
            const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)tri_pk;
            r0 = pk[0]; r1 = pk[1]; r2 = pk[2];
        }
        if (want_node) {
            const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending;
            n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4];
        }
        if (has_tri) {
            float t, U, V, ad;
            if (tri_test(r0, r1, r2, o, d, 0.0f, best_t, t, U, V, ad)) {
                bool better = t < best_t || best_pk == 0xFFFFFFFFu;
                if (!better) better = (MRT_LANE_HIT_LDS && hitw) ? __float_as_uint(r0.w) < __float_as_uint(hitw[192u + lane]) : __float_as_uint(r0.w) < __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);      // t == best_t: ties go to the lowest id
                if (better) { best_t = t; best_pk = tri_pk; if (MRT_LANE_HIT_LDS && hitw) { hitw[lane] = U; hitw[64u + lane] = V; hitw[128u + lane] = ad; hitw[192u + lane] = r0.w; } }
            }
        }
        if (want_node) {
            uint32_t node_hits, tri_hits;
            wide_node_test<MRT_WIDE_SCALED != 0>(n0, n1, n2, n3, n4, o, ix, iy, iz, nx, ny, nz, oct, 0.0f, best_t, node_hits, tri_hits);
            uint32_t sp = g_mask >> 16;
            if ((g_mask & 0xFF00u) != 0) { wstack_push(stack, sp, lane, g_base, g_mask & 0xFFFFu); sp++; }     // siblings still to visit
            g_base = __float_as_uint(n1.x); g_mask = (sp << 16) | (node_hits << 8) | (__float_as_uint(n0.w) >> 24);
            t_base = __float_as_uint(n1.y); t_mask = tri_hits;
        }
        else if (t_rest == 0u) break;        // no node left and this was the last pending triangle
    }
    h.t = best_t; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu; h.pk = best_pk;
    if (best_pk == 0xFFFFFFFFu) return false;
    if (MRT_LANE_HIT_LDS && hitw) { h.U = hitw[lane]; h.V = hitw[64u + lane]; h.ad = hitw[128u + lane]; h.gid = __float_as_uint(hitw[192u + lane]); return true; }
    // id and barycentrics of the winning triangle: recomputed (same arithmetic) instead of living in four registers through the loop
    const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)best_pk;
    const float4 q0 = pk[0];
    float t_;
    (void)tri_test(q0, pk[1], pk[2], o, d, 0.0f, __builtin_inff(), t_, h.U, h.V, h.ad);
    h.gid = __float_as_uint(q0.w);
    return true;
}

// The same for TWO-LEVEL scenes (wnodes = [8-wide TLAS | BLASes], see traverse_wide_stream<TWO_LEVEL> below): one ray per lane walks both levels on one stack — at an instance it
// parks the TLAS group, goes into object space (direction not renormalised: t stays the world distance), walks the BLAS and comes back.  The primary rays of a tile enter and
// leave the same instances together, so the lanes stay in step across the level changes that put the stream loop's incoherent rays out of step.  The world ray is the caller's
// (wo, wd: the camera position and the primary direction, live in registers anyway): nothing is parked in LDS.  g_mask: imask | hit bits << 8 | stack depth << 16 | entry depth << 24.
// seed (SEED): (packet | instance << 24) already tested by the caller, its distance in tmax.  h.pk = the final hit in the same encoding; h.gid = the GLOBAL triangle id.
template <bool SEED>
MRT_DEV bool traverse_wide_lane_two_level(const SceneView &s, const f3 wo, const f3 wd, float tmax, uint32_t seed, TravHit &h, uint32_t *stack) {
    const uint32_t lane = threadIdx.x & 63;
    f3 o = wo, d = wd;
    float ix = box_inv(d.x), iy = box_inv(d.y), iz = box_inv(d.z);
    bool nx = d.x < 0.0f, ny = d.y < 0.0f, nz = d.z < 0.0f;
    uint32_t oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
    float best_t = tmax; uint32_t best_pk = 0xFFFFFFFFu;
    uint32_t insts = 0;                                   // instance being walked | instance of the closest hit << 16
    if (SEED && seed != 0xFFFFFFFFu) { best_pk = seed & 0xFFFFFFu; insts = (seed >> 24) << 16; }
    uint32_t tl_pack = 0;                                 // the TLAS leaf's remaining instances while inside a BLAS: tri_base << 8 | mask
    uint32_t g_base = 0, g_mask = s.num_wnodes != 0 ? 0x100u : 0u, t_base = 0, t_mask = 0;
    for (;;) {
        {   // nothing of the BLAS left (no triangle pending, no hit child, stack back at the entry depth): back to world space and to the TLAS group parked at entry
            const uint32_t sp = (g_mask >> 16) & 0xFFu, isp = g_mask >> 24;
            if (isp != 0u && t_mask == 0u && (g_mask & 0xFF00u) == 0u && sp == isp) {
                o = wo; d = wd; ix = box_inv(d.x); iy = box_inv(d.y); iz = box_inv(d.z);
                nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f; oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                wstack_pop(stack, sp - 1u, lane, g_base, g_mask); g_mask |= (sp - 1u) << 16;
                t_base = tl_pack >> 8; t_mask = tl_pack & 0xFFu;
            }
        }
        const bool in_blas = (g_mask >> 24) != 0u;
        const bool has_inst = t_mask != 0u && !in_blas;          // at the TLAS level a pending "triangle" is an instance to enter
        bool has_tri = t_mask != 0u && !has_inst;
        uint32_t t_rest = t_mask & (t_mask - 1u);
        bool want_node = t_rest == 0u && !has_inst;
        uint32_t pending = 0, tri_pk = 0;
        if (has_inst) {
            const uint32_t k = (uint32_t)__ffs((int)t_mask) - 1u;
            t_mask &= t_mask - 1u;
            const uint32_t id = s.wtlas_index[t_base + k];
            const InstanceDev &I = s.inst[id];
            tl_pack = (t_base << 8) | t_mask;
            uint32_t sp = (g_mask >> 16) & 0xFFu;
            wstack_push(stack, sp, lane, g_base, g_mask & 0xFFFFu); sp++;          // parked always, also without siblings left: the exit pops it
            const f3 oo = to_object_point(I, o), dd = to_object_dir(I, d);
            o = oo; d = dd; ix = box_inv(d.x); iy = box_inv(d.y); iz = box_inv(d.z);
            nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f; oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
            g_base = 0; g_mask = (sp << 24) | (sp << 16);
            t_base = 0; t_mask = 0;
            insts = (insts & 0xFFFF0000u) | id;
            if (I.ntri <= 8u) { t_base = I.packet_base; t_mask = (1u << I.ntri) - 1u; t_rest = t_mask & (t_mask - 1u); has_tri = true; }      // a wall, the floor: its packets are the pending set
            else { pending = I.wroot; want_node = true; }
        }
        else if (want_node) {
            if ((g_mask & 0xFF00u) == 0) {
                const uint32_t sp = (g_mask >> 16) & 0xFFu, isp = g_mask >> 24;
                if (isp != 0u && sp == isp) want_node = false;          // the BLAS's last triangle is tested in this iteration; the lane leaves in the next
                else if (sp == 0) { want_node = false; if (!has_tri) break; }
                else { wstack_pop(stack, sp - 1u, lane, g_base, g_mask); g_mask |= ((sp - 1u) << 16) | (isp << 24); }
            }
            if (want_node) {
                const uint32_t hits = (g_mask >> 8) & 0xFFu;
                const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;
                g_mask &= ~(0x100u << b);
                const uint32_t slot = b ^ oct;
                pending = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));
            }
        }
        float4 r0, r1, r2, n0, n1, n2, n3, n4;
        asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r2.x), "=v"(r2.y), "=v"(r2.z));
        asm volatile("" : "=v"(n0.x), "=v"(n0.y), "=v"(n0.z), "=v"(n0.w), "=v"(n1.x), "=v"(n1.y), "=v"(n1.z), "=v"(n1.w), "=v"(n2.x), "=v"(n2.y), "=v"(n2.z), "=v"(n2.w));
        asm volatile("" : "=v"(n3.x), "=v"(n3.y), "=v"(n3.z), "=v"(n3.w), "=v"(n4.x), "=v"(n4.y), "=v"(n4.z), "=v"(n4.w));
        r1.w = 0.0f; r2.w = 0.0f;
        if (has_tri) {
            tri_pk = t_base + (uint32_t)__ffs((int)t_mask) - 1u; t_mask = t_rest;
            const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)tri_pk;
            r0 = pk[0]; r1 = pk[1]; r2 = pk[2];
        }
        if (want_node) {
            const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending;
            n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4];
        }
        if (has_tri) {
            float t, U, V, ad;
            if (tri_test(r0, r1, r2, o, d, 0.0f, best_t, t, U, V, ad)) {
                bool better = t < best_t || best_pk == 0xFFFFFFFFu;
                if (!better) better = s.inst[insts & 0xFFFFu].gid_base + __float_as_uint(r0.w) < s.inst[insts >> 16].gid_base + __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);      // t == best_t: lowest global id
                if (better) { best_t = t; best_pk = tri_pk; insts = (insts & 0xFFFFu) | (insts << 16); }
            }
        }
        if (want_node) {
            uint32_t node_hits, tri_hits;
            wide_node_test<MRT_WIDE_SCALED != 0>(n0, n1, n2, n3, n4, o, ix, iy, iz, nx, ny, nz, oct, 0.0f, best_t, node_hits, tri_hits);
            uint32_t sp = (g_mask >> 16) & 0xFFu;
            const uint32_t isp = g_mask & 0xFF000000u;
            if ((g_mask & 0xFF00u) != 0) { wstack_push(stack, sp, lane, g_base, g_mask & 0xFFFFu); sp++; }
            g_base = __float_as_uint(n1.x); g_mask = isp | (sp << 16) | (node_hits << 8) | (__float_as_uint(n0.w) >> 24);
            t_base = __float_as_uint(n1.y); t_mask = tri_hits;
        }
    }
    h.t = best_t; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu; h.pk = 0xFFFFFFFFu;
    if (best_pk == 0xFFFFFFFFu) return false;
    // id and barycentrics of the winning triangle, in the object space of its instance (the walk's own arithmetic)
    const InstanceDev &I = s.inst[insts >> 16];
    const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)best_pk;
    const float4 q0 = pk[0];
    float t_;
    (void)tri_test(q0, pk[1], pk[2], to_object_point(I, wo), to_object_dir(I, wd), 0.0f, __builtin_inff(), t_, h.U, h.V, h.ad);
    h.gid = I.gid_base + __float_as_uint(q0.w);
    h.pk = best_pk | ((insts >> 16) << 24);
    return true;
}

// ---------------------------------------------------------------------------------------------
// Wave-level stream traversal with lane refill (the "persistent wavefront + ray compaction" of BASELINE.json
// configs[4]) on the wide layout.  rocprofv3 on the one-ray-per-lane kernel: VALUBusy 81 %, VALUUtilization 19 % —
// the VALU is saturated by instructions in which four lanes out of five are idle, because the rays of a wave
// finish at very different times.  Here a wave owns a range of consecutive rays; the next 64 of them are
// PREFETCHED into registers (all 64 lanes, coalesced), and whenever WIDE_REFILL_AT lanes are idle they take the
// next prefetched rays by cross-lane reads (ds_bpermute), without touching memory.  A refilled lane starts with
// an empty stack, so nothing but the ray moves.
//
//   fetch(i, A, B, tag, is_any)   ray i of the wave's range: A = origin | tmax, B.xyz = direction, tag < 2^31 (caller's id)
//   emit(tag, is_any, hit, h)     result of a finished ray; called by all idle lanes together at refill time, so that
//                                 its divisions and stores do not run at one or two lanes in the iteration a ray ends in
//
// Register budget: 72 VGPRs = 7 waves per SIMD (MRT_WIDE_STREAM_WAVES; the two-level instantiation: 77-80 VGPRs, 6 waves); the one-round-trip
// iteration below holds a node (20 registers) and a packet (10) at once.
#ifndef MRT_WIDE_REFILL_AT
#define MRT_WIDE_REFILL_AT 16
#endif
// waves per SIMD the stream kernels are compiled for
#ifndef MRT_TWO_LEVEL_WAVES
#define MRT_TWO_LEVEL_WAVES 6
#endif
#ifndef MRT_WIDE_STREAM_WAVES
#define MRT_WIDE_STREAM_WAVES (MRT_WIDE_SPEC ? 6 : 7)      // the second triangle group costs two registers: 80 instead of 72 (no spills); the frame rate does not depend on 6 or 7 waves per SIMD (docs/HISTORY.md §6)
#endif
constexpr int WIDE_REFILL_AT = MRT_WIDE_REFILL_AT;

struct StreamStats {
    uint32_t iters, live_sum, node_sum, tri_sum, refills, refill_lanes;
#ifdef MRT_STATS_PROBE      // diagnostics builds (tools/build_variant.sh; tools/stream_level_probe.py): = 1 node fetches by tree level, = 2 pending triangles per iteration (what pooling triangle tests across lanes could use)
    uint32_t level_end[3] = {0, 0, 0};      // = 1: first node index beyond levels 0..1, 0..2, 0..3 (BFS numbering); the three counters below: fetches of nodes before each
    uint32_t probe[3] = {0, 0, 0};          // = 2: pending triangles summed over lanes and iterations | lanes with two or more pending | iterations in which fewer than 16 lanes test a triangle
#endif
#ifdef MRT_WAVE_TIMES
    uint32_t drain_iters = 0, drain_live = 0, maxdt = 0, drain_le8 = 0; unsigned long long prev = 0ull, drain_t0 = 0ull;
    // what the last live lanes of a draining wave still hold (could an idle lane take a subtree off them? round 6): over the drain iterations with at most 16 ([0]) / at most 4 ([1])
    // live lanes, per live lane: histogram of its stack depth (0, 1, 2, 3, 4, >= 5 parked sibling groups), the hit children still to visit in those groups and in its current one, samples
    uint32_t dr_hist[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}}, dr_kids[2] = {0, 0}, dr_n[2] = {0, 0}, dr_tris[2] = {0, 0};
#endif
};

// A wave's supply of rays: consecutive chunks [b, e) of the queue.  `next(b, e)` is called by the whole wave (wave-uniform
// result) when the current chunk is used up and returns false when there is nothing left.
struct OneRange {            // the static split: the wave owns one range
    uint32_t b, e; bool used = false;
    MRT_DEV bool operator()(uint32_t &ob, uint32_t &oe) { if (used) return false; used = true; ob = b; oe = e; return b < e; }
};
// The static split in 64-ray batches: `step` = 64 walks one contiguous range [next, limit) like OneRange; a larger step deals the queue's batches round-robin to the launch's
// waves (renderer option stream_stride): neighbouring rays cost alike, so a contiguous range makes some waves several times longer than the average one.
struct BatchStride {
    uint32_t next, step, limit;
    MRT_DEV bool operator()(uint32_t &ob, uint32_t &oe) { if (next >= limit) return false; ob = next; oe = min(limit, next + 64u); next += step; return true; }
};
// The dynamic split ("persistent waves"): every resident wave keeps pulling `chunk` rays from a shared counter until the queue is
// empty, so all waves of a launch end within one chunk of each other instead of the last round of a static grid running on a
// half-empty chip (rocprofv3, serialised 4-frame launches: 16 K static waves of 1024 rays on 7 168 wave slots held 47 % of them
// on average).  One returning atomic per chunk: a counter word sustains ~88 of them per microsecond (MI355X_MICROARCH.md "dequeue").
struct SharedCounter {
    uint32_t *counter; uint32_t n, chunk;
    MRT_DEV bool operator()(uint32_t &ob, uint32_t &oe) {
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, chunk);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n) return false;
        ob = base; oe = min(n, base + chunk);
        return true;
    }
};

// The dynamic split with one counter per XCD.  256 CUs pulling from ONE word queue up behind it (a word takes ~88 returning atomics per microsecond, and a pull under that load
// takes 1.4-3 us: MI355X_MICROARCH.md "dequeue": shard the head per XCD above 64 pullers).  Here the queue is cut into eight REGIONS, one per XCD (the workgroups that share
// blockIdx.x % 8 share an XCD: they are dealt round-robin), each with its own counter on its own cache line.  Region x is the x-th eighth of every sub-frame's share of the queue
// — a pass's queue is filled sub-frame after sub-frame, in screen-tile order within each, by blocks that reserve their range when they finish — so an XCD walks rays that left
// one band of the screen (measured: the L2 miss count does not change with it, what is gained is the counter).  The combined queue [bounce rays | shadow rays] is cut the same
// way in each part.  A wave whose home region is used up goes on with the next one (x + 1, ...): all regions are emptied whatever the number of waves, and every counter only
// ever grows, so a wave asks at most (chunks + 8) times.  subframes == 0: one region over the whole combined queue (the form above, counting chunks instead of rays).
constexpr uint32_t XCD_COUNTER_STRIDE = 32;            // words between two regions' counters: 128 bytes
struct XcdRegions {
    uint32_t *counters; uint32_t n_next, n, chunk, subframes; uint32_t home; uint32_t tries = 0;
    MRT_DEV bool operator()(uint32_t &ob, uint32_t &oe) {
        const uint32_t regions = subframes ? 8u : 1u;
        const uint32_t C1 = subframes ? (n_next + chunk - 1u) / chunk : (n + chunk - 1u) / chunk, C2 = subframes ? (n - n_next + chunk - 1u) / chunk : 0u;      // chunks of the two parts
        const uint32_t B = subframes ? subframes : 1u;
        const uint32_t F1 = (C1 + B - 1u) / B, S1 = (F1 + regions - 1u) / regions, F2 = (C2 + B - 1u) / B, S2 = (F2 + regions - 1u) / regions;      // chunks per sub-frame, and per region of a sub-frame
        const uint32_t K1 = B * S1, K = K1 + B * S2;       // a region's list: its slices of part 1, sub-frame after sub-frame, then of part 2
        while (tries < regions) {
            const uint32_t x = (home + tries) & (regions - 1u);
            uint32_t k = 0;
            if ((threadIdx.x & 63) == 0) k = atomicAdd(&counters[x * XCD_COUNTER_STRIDE], 1u);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            if (k >= K) { tries++; continue; }
            const bool p2 = k >= K1;
            const uint32_t kk = p2 ? k - K1 : k, S = p2 ? S2 : S1, F = p2 ? F2 : F1, Cp = p2 ? C2 : C1;
            const uint32_t f = kk / S, lc = x * S + (kk - f * S);
            if (lc >= F) continue;                            // (rounding: the last region's slice is shorter)
            const uint32_t c = f * F + lc;
            if (c >= Cp) continue;                            // (rounding: the last sub-frame's share is shorter)
            const uint32_t first = subframes && p2 ? n_next : 0u, lim = subframes && !p2 ? n_next : n;
            ob = first + c * chunk; oe = min(lim, ob + chunk);
            return true;
        }
        return false;
    }
};

// Two-level scenes, binned form (renderer option tl_pairs).  The TLAS pass — this loop with a PairQueue — tests instances of at most eight triangles in place and,
// instead of entering a larger one, appends {ray, instance} to a queue; a second launch walks every pair in object space with the FLATTENED loop (TWO_LEVEL = false, per-ray root:
// ROOTS), its lanes never changing level, and folds the hits into the rays' results with atomics.  When the queue is full a lane enters the instance in place, as without a queue.
struct NoPairs { static constexpr bool on = false; };
struct PairQueue {
    static constexpr bool on = true;
    static constexpr uint32_t WORDS = 1u;             // float4 per pair: {ray, instance, bound, report tag}
    static constexpr uint32_t BLOCK = 256;            // pair slots a wave reserves at a time: ONE atomic on the queue's counter per 256 pairs (a counter word sustains ~88 returning
                                                      // atomics per microsecond; one per wave and iteration — the first form — made the TLAS pass five times slower than the walk it replaces)
    uint4 *__restrict__ pairs; uint32_t *__restrict__ count; uint32_t cap;
    uint32_t *cursor;                                 // LDS, two words of this wave: {next free slot, end of the wave's block}; both 0 at the start
    // called by the lanes that reached a large instance in this iteration (a divergent branch: the ballot sees exactly them); true = the pair is stored
    MRT_DEV bool push(uint32_t ray_index, uint32_t inst, float tmax, uint32_t tagw) const {
        const unsigned long long m = __ballot(1);
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t lane = threadIdx.x & 63, n = (uint32_t)__popcll(m), rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        uint32_t cur = 0, fits = 0, fresh = 0xFFFFFFFFu;          // leader: the old block's cursor, how many of the n pairs it still takes, the new block (none / refused: ~0)
        if ((int)lane == leader) {
            cur = cursor[0]; const uint32_t end = cursor[1];
            fits = min(n, end - cur);
            if (fits < n) {
                const uint32_t base = atomicAdd(count, BLOCK);
                if (base < cap && cap - base >= BLOCK) { fresh = base; cursor[0] = base + (n - fits); cursor[1] = base + BLOCK; }
                else { cursor[0] = end; }                           // the queue is full: the lanes beyond `fits` walk their instance in place
            } else cursor[0] = cur + n;
        }
        cur = (uint32_t)__builtin_amdgcn_readlane((int)cur, leader); fits = (uint32_t)__builtin_amdgcn_readlane((int)fits, leader); fresh = (uint32_t)__builtin_amdgcn_readlane((int)fresh, leader);
        uint32_t slot;
        if (rank < fits) slot = cur + rank;
        else if (fresh != 0xFFFFFFFFu) slot = fresh + (rank - fits);
        else return false;
        qstore(reinterpret_cast<float4 *>(&pairs[slot]), make_float4(__uint_as_float(ray_index), __uint_as_float(inst), tmax, __uint_as_float(tagw)));      // (written once, read once by the next launch)
        return true;
    }
    // the whole wave, when it has no rays left: the unused rest of its block becomes pairs that name no ray (the BLAS pass skips them)
    MRT_DEV void close() const {
        const uint32_t lane = threadIdx.x & 63, cur = cursor[0], end = cursor[1];
        for (uint32_t k = cur + lane; k < end; k += 64) qstore(reinterpret_cast<float4 *>(pairs) + k, make_float4(__uint_as_float(0xFFFFFFFFu), 0.0f, 0.0f, 0.0f));
        cursor[0] = end;
    }
};

// Optional LDS extra of the stream walk (flattened scenes; renderer option hit_lds, docs/HISTORY.md round 5): the lane's closest hit so far keeps its U, V, |det| and
// triangle id in four words of LDS, written when a closer hit is found, so that a finished ray is reported from LDS instead of fetching the winning packet again and
// re-running its triangle test at refill time (a dependent round trip and ~100 VALU instructions at a quarter of the lanes, three to four times per 64 rays).
// (Staging the top levels of the tree in LDS beside it — BASELINE.json's "LDS-staged BVH nodelets" — measured -5 ... -8 % in four forms and was removed in round 6:
// those nodes already hit L1 and the kernel is issue-bound; profiles/r05_lds_top_ab.txt, docs/HISTORY.md.)
struct NoExt { static constexpr bool hit_lds = false, root_pre = false; float *hit = nullptr; uint32_t *root = nullptr; };      // (members never read: every use sits behind a flag)
template <bool HIT> struct StreamExt {
    static constexpr bool hit_lds = HIT, root_pre = HIT && MRT_ROOT_AT_FETCH != 0;
    float *hit;             // HIT: [4][64] words of this wave: U, V, |det|, triangle id (bits) of the lane's closest hit so far
    uint32_t *root = nullptr;      // MRT_ROOT_AT_FETCH: [2][64] words of this wave: the root's {internal-child hits, leaf-triangle hits} of the prefetched batch's rays
};
constexpr uint32_t HIT_LDS_WORDS = 256u + (MRT_ROOT_AT_FETCH ? 128u : 0u);      // per wave, in front of its stack
//
// TWO_LEVEL (scenes committed with instancing = 1, two_level.hip): wnodes[0 ..] is an 8-wide TLAS whose leaf children are single instances
// (the "packet" tri_base + k is an entry of wtlas_index), followed by the BLASes' nodes with absolute indices.  The same loop walks both levels on
// the same stack: a lane whose pending "triangle" is an instance parks the TLAS group it came from on the stack, takes its ray into object space
// (direction not renormalised: t stays the world-space distance) and enters the BLAS root; when nothing of the BLAS is left — stack depth back
// at the entry depth — it restores the world ray from LDS and pops the parked group.  g_mask carries the entry depth in bits 24..28 (0 = at the
// TLAS level), the stack depth in bits 16..20.  `stack` then starts with WIDE_WORLD_RAY_BYTES of parked world rays.
// SEED (primary rays with a hint, k_trace_primary_wide_stream): `fetch` also returns a candidate hit — a packet (| instance << 24) whose distance it has already put
// into the ray's limit word — and the walk starts with it as its closest hit so far; TravHit::pk at emit time is the final hit in the same encoding.
// ROOTS (with TWO_LEVEL = false): `fetch` also returns the node the ray's walk starts at (the root of its instance's BLAS in the shared node array) instead of node 0.
template <bool TWO_LEVEL = false, bool SEED = false, bool ROOTS = false, class Pairs = NoPairs, class Ext = NoExt, class Chunks, class RayFetch, class Emit>
MRT_DEV void traverse_wide_stream(const SceneView &s, Chunks next_chunk, uint32_t *stack, RayFetch fetch, Emit emit, StreamStats *ss = nullptr, Pairs pq = Pairs{}, Ext ext = Ext{}) {
    static_assert(!Ext::hit_lds || (!TWO_LEVEL && !SEED), "hit_lds: flattened scenes, rays without a seed hit");
    const uint32_t lane = threadIdx.x & 63;
    float *const wray = reinterpret_cast<float *>(stack);      // [6][64]: o.xyz, d.xyz of the lane's ray in world space (TWO_LEVEL)
    if (TWO_LEVEL) stack += WIDE_WORLD_RAY_BYTES / 4u;
    uint32_t insts = 0;                                   // TWO_LEVEL: instance being walked | instance of the closest hit << 16
    uint32_t tl_pack = 0;                                 // TWO_LEVEL: the TLAS leaf's remaining instances while inside a BLAS: tri_base << 8 | mask
    const unsigned long long lt = (1ull << lane) - 1ull;
    // prefetched batch: rays batch_base .. batch_base + batch_n - 1, one per lane; pB.w = tag | any-hit flag << 31
    float4 pA = make_float4(0, 0, 0, 0), pB = pA;
    uint32_t pS = 0xFFFFFFFFu;                        // SEED: the prefetched ray's candidate; ROOTS: its start node
    uint32_t batch_n = 0, batch_used = 0;             // wave-uniform; used == n -> nothing prefetched
    uint32_t batch_first = 0, qi = 0;                 // Pairs: index of the batch's first ray in the launch's queue, and of the lane's own ray
    bool pq_refused = false;                          // Pairs: the queue refused one of this lane's pairs (it is full)
    uint32_t cur = 0, end = 0;                        // unfetched part of the current chunk
    bool more = true;                                 // the chunk source may have more
    bool draining = false;                            // nothing left to hand out: the wave runs until its last rays are done
    // live ray
    bool live = false, unreported = false;            // unreported: the lane's ray is finished, its result not yet emitted
    uint32_t tagw = 0;                                // tag | any-hit flag << 31
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1); float ix = 0, iy = 0, iz = 0; bool nx = false, ny = false, nz = false; uint32_t oct = 0;
    float best_t = 0.0f; uint32_t best_pk = 0xFFFFFFFFu;     // closest hit so far: distance and packet (0xFFFFFFFF = none); id, U, V are re-read at emit time
    uint32_t g_base = 0, g_mask = 0, t_base = 0, t_mask = 0;   // g_mask: imask | permuted hit bits << 8 | stack depth << 16
    // Node visits ahead of the triangle tests (flattened scenes).  A node's leaf children leave up to 32 pending triangles, tested one per iteration; a lane
    // used to visit its next node only together with the LAST of them, so that in an average iteration half the live lanes sat out the node test — the
    // expensive part of the iteration, paid by the whole wave (rocprofv3: 37 % of the lanes active per VALU instruction).  Now a lane with pending triangles
    // visits its next node in the same iteration and parks that node's triangles in a second group (u_base, u_mask); it sits out only while both groups
    // are occupied.  The visit sees a limit the pending triangles may still shorten — more children accepted than strictly needed, never fewer — and the
    // closest hit stays the minimum over (t, id) whatever the order: the image is unchanged.  A ray then needs about max(node visits, triangle tests)
    // iterations instead of their sum.
    // (two-level scenes: tried inside the BLASes, MRT_WIDE_SPEC_TWO_LEVEL — 86 instead of 80 registers and no faster, twice: 6.27 vs 6.38 and 6.65 vs 6.63 Grays/s on dragon x 4; left off there)
    constexpr bool SPEC = (TWO_LEVEL ? MRT_WIDE_SPEC_TWO_LEVEL : MRT_WIDE_SPEC) != 0;
    uint32_t u_base = 0, u_mask = 0;
    for (;;) {
        const unsigned long long m_idle = __ballot(!live);
        const uint32_t n_idle = (uint32_t)__popcll(m_idle);
        if (n_idle >= (uint32_t)WIDE_REFILL_AT || m_idle == ~0ull) {
            if (unreported) {
                const bool was_any = (tagw >> 31) != 0, was_hit = best_pk != 0xFFFFFFFFu;
                TravHit h; h.t = best_t; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu;
                if constexpr (Ext::hit_lds) {
                    if (!was_any && was_hit) { h.U = ext.hit[lane]; h.V = ext.hit[64u + lane]; h.ad = ext.hit[128u + lane]; h.gid = __float_as_uint(ext.hit[192u + lane]); }      // kept since the hit was found
                }
                else if (!was_any && was_hit) {      // id and barycentrics of the winning triangle: recomputed here (same arithmetic) instead of living in 4 registers
                    const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)best_pk;
                    const float4 q0 = pk[0];
                    float t_;
                    if (TWO_LEVEL) {            // in the object space of the hit's instance, from the parked world ray
                        const InstanceDev &I = s.inst[insts >> 16];
                        const f3 wo = mk3(wray[lane], wray[64 + lane], wray[128 + lane]), wd = mk3(wray[192 + lane], wray[256 + lane], wray[320 + lane]);
                        (void)tri_test(q0, pk[1], pk[2], to_object_point(I, wo), to_object_dir(I, wd), 0.0f, __builtin_inff(), t_, h.U, h.V, h.ad);
                        h.gid = I.gid_base + __float_as_uint(q0.w);
                    } else {
                        (void)tri_test(q0, pk[1], pk[2], o, d, 0.0f, __builtin_inff(), t_, h.U, h.V, h.ad);
                        h.gid = __float_as_uint(q0.w);
                    }
                }
                if (SEED) h.pk = was_hit ? (best_pk | (TWO_LEVEL ? (insts >> 16) << 24 : 0u)) : 0xFFFFFFFFu;
                emit(tagw & 0x7FFFFFFFu, was_any, was_hit, h); unreported = false;
            }
            if (batch_used >= batch_n) {                        // prefetch the next (up to) 64 rays (coalesced), all lanes
                if (cur >= end && more) more = next_chunk(cur, end);
                batch_n = cur < end ? min(64u, end - cur) : 0u; batch_used = 0;
                if (Pairs::on) batch_first = cur;
                if (lane < batch_n) {
                    uint32_t tag = 0, is_any = 0;
                    if constexpr (SEED || ROOTS) fetch(cur + lane, pA, pB, tag, is_any, pS); else fetch(cur + lane, pA, pB, tag, is_any);
                    pB.w = __uint_as_float((tag & 0x7FFFFFFFu) | (is_any << 31));
                }
                if constexpr (Ext::root_pre) {
                    if (batch_n != 0u && s.num_wnodes != 0) {          // (wave-uniform) the root against every ray of the batch, here
                        const float4 *__restrict__ nd = s.wnodes;
                        const float4 r0_ = nd[0], r1_ = nd[1], r2_ = nd[2], r3_ = nd[3], r4_ = nd[4];
                        if (lane < batch_n) {
                            const f3 o_ = mk3(pA.x, pA.y, pA.z);
                            const float ix_ = box_inv(pB.x), iy_ = box_inv(pB.y), iz_ = box_inv(pB.z);
                            const bool nx_ = pB.x < 0.0f, ny_ = pB.y < 0.0f, nz_ = pB.z < 0.0f;
                            uint32_t nh_, th_;
                            wide_node_test<MRT_WIDE_SCALED != 0>(r0_, r1_, r2_, r3_, r4_, o_, ix_, iy_, iz_, nx_, ny_, nz_, (nx_ ? 1u : 0u) | (ny_ ? 2u : 0u) | (nz_ ? 4u : 0u), 0.0f, pA.w, nh_, th_);
                            ext.root[lane] = nh_; ext.root[64u + lane] = th_;
                        }
                    }
                }
                cur += batch_n;
            }
            const uint32_t avail = batch_n - batch_used;
            if (avail == 0) { if (m_idle == ~0ull) break; draining = true; }
            else {
                const uint32_t rank = (uint32_t)__popcll(m_idle & lt);
                const bool take = !live && rank < avail;
                // cross-lane reads are executed by every lane (bpermute needs the whole wave)
                const int sl = (int)(take ? batch_used + rank : lane);
                const float ax_ = __shfl(pA.x, sl), ay_ = __shfl(pA.y, sl), az_ = __shfl(pA.z, sl), aw_ = __shfl(pA.w, sl);
                const float bx_ = __shfl(pB.x, sl), by_ = __shfl(pB.y, sl), bz_ = __shfl(pB.z, sl), bw_ = __shfl(pB.w, sl);
                const uint32_t seed_ = (SEED || ROOTS) ? (uint32_t)__shfl((int)pS, sl) : 0xFFFFFFFFu;
                if (take) {
                    o = mk3(ax_, ay_, az_); d = mk3(bx_, by_, bz_); ix = box_inv(bx_); iy = box_inv(by_); iz = box_inv(bz_);
                    nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f; oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                    best_t = aw_; best_pk = 0xFFFFFFFFu;
                    tagw = __float_as_uint(bw_);
                    // enter the root as the only "hit child" of a pseudo group: base 0, no internal-child bits -> node 0; empty stack
                    g_base = ROOTS ? seed_ : 0u; g_mask = s.num_wnodes != 0 ? 0x100u : 0u; t_base = 0; t_mask = 0;      // (the pseudo group has no internal-child bits: its one "hit child" is node g_base itself)
                    if constexpr (Ext::root_pre) {
                        if (s.num_wnodes != 0) {          // the root was tested when the batch was fetched: its children are the lane's first group
                            const float4 q1_ = s.wnodes[1];
                            g_base = __float_as_uint(q1_.x); g_mask = (ext.root[sl] << 8) | (__float_as_uint(s.wnodes[0].w) >> 24);
                            t_base = __float_as_uint(q1_.y); t_mask = ext.root[64u + (uint32_t)sl];
                        }
                    }
                    if (SPEC) { u_base = 0; u_mask = 0; }
                    live = true;
                    if (Pairs::on) qi = batch_first + (uint32_t)sl;
                    if (TWO_LEVEL) {
                        wray[lane] = ax_; wray[64 + lane] = ay_; wray[128 + lane] = az_; wray[192 + lane] = bx_; wray[256 + lane] = by_; wray[320 + lane] = bz_;
                        insts = 0; tl_pack = 0;
                    }
                    if (SEED && seed_ != 0xFFFFFFFFu) { best_pk = TWO_LEVEL ? (seed_ & 0xFFFFFFu) : seed_; if (TWO_LEVEL) insts = (seed_ >> 24) << 16; }
                }
                batch_used += min(avail, n_idle);
#ifndef MRT_STATS_BOTH
                if (ss) { ss->refills++; ss->refill_lanes += min(avail, n_idle); }
#endif
                continue;
            }
        }
#ifdef MRT_WAVE_TIMES
        if (ss) {
            if (draining) {
                const uint32_t nl_ = (uint32_t)__popcll(__ballot(live)); ss->drain_iters++; ss->drain_live += nl_; if (nl_ <= 8u) ss->drain_le8++;
                if (!TWO_LEVEL && nl_ <= 16u) {          // (flattened scenes) every live lane's pending work, summed over the wave by ballots / shuffles; lane 0's copy of `ss` is the one reported
                    const uint32_t sp_ = live ? g_mask >> 16 : 0u;
                    uint32_t kids_ = live ? (uint32_t)__popc((g_mask >> 8) & 0xFFu) : 0u;
                    for (uint32_t l_ = 0; l_ < sp_; l_++) kids_ += (uint32_t)__popc(stack[l_ * (WIDE_STACK_LEVEL_BYTES / 4u) + lane] & 0xFFu);
                    uint32_t tris_ = live ? (uint32_t)__popc(t_mask) + (uint32_t)__popc(u_mask) : 0u;
                    for (int o_ = 32; o_ > 0; o_ >>= 1) { kids_ += (uint32_t)__shfl_xor((int)kids_, o_); tris_ += (uint32_t)__shfl_xor((int)tris_, o_); }
                    for (int c_ = 0; c_ < 2; c_++) if (c_ == 0 || nl_ <= 4u) {
                        for (uint32_t d_ = 0; d_ < 6u; d_++) ss->dr_hist[c_][d_] += (uint32_t)__popcll(__ballot(live && (d_ < 5u ? sp_ == d_ : sp_ >= 5u)));
                        ss->dr_kids[c_] += kids_; ss->dr_tris[c_] += tris_; ss->dr_n[c_] += nl_;
                    }
                }
            }
            const unsigned long long now_ = wall_clock64(); const uint32_t dt_ = (uint32_t)(now_ - ss->prev);
            if (ss->prev != 0ull && dt_ > ss->maxdt) ss->maxdt = dt_;
            ss->prev = now_; if (draining && ss->drain_t0 == 0ull) ss->drain_t0 = now_;
        }
#endif
        if (ss) { ss->iters++; ss->live_sum += (uint32_t)__popcll(__ballot(live)); }
        // One memory round trip per iteration.  A lane with at most one triangle left to test already knows the next node
        // it will visit (the nearest remaining hit child, or the top of its stack), so it fetches that node (80 B) together
        // with the triangle packet (48 B), tests the triangle, then the node's eight boxes against the possibly shorter ray.
        // (Measured, full frame: node-or-triangle per iteration 6.90, node then triangle with two round trips 7.24,
        // this loop 7.44 Grays/s.)
        if (SPEC && t_mask == 0u && u_mask != 0u) { t_base = u_base; t_mask = u_mask; u_mask = 0u; }      // the first group is used up: the second takes its place
        if (TWO_LEVEL) {
            // a lane inside a BLAS with nothing of it left (no triangle pending, no hit child, stack back at the entry depth) returns to world
            // space and to the TLAS group parked at entry — in this same iteration it goes on to its next instance or TLAS node
            const uint32_t sp = (g_mask >> 16) & 0xFFu, isp = g_mask >> 24;
            if (live && isp != 0u && t_mask == 0u && (g_mask & 0xFF00u) == 0u && sp == isp) {
                o = mk3(wray[lane], wray[64 + lane], wray[128 + lane]); d = mk3(wray[192 + lane], wray[256 + lane], wray[320 + lane]);
                ix = box_inv(d.x); iy = box_inv(d.y); iz = box_inv(d.z);
                nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f; oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                wstack_pop(stack, sp - 1u, lane, g_base, g_mask); g_mask |= (sp - 1u) << 16;
                t_base = tl_pack >> 8; t_mask = tl_pack & 0xFFu;
            }
        }
        const bool in_blas = TWO_LEVEL && (g_mask >> 24) != 0u;
        // Drain phase (rocprofv3 / tools/archive/wave_times.py: the last 30 % of a launch run on < 2 % of the waves, each walking one or two grazing rays
        // that test 100-200 triangles one per iteration).  The finished lanes help: the pending triangles of ONE such ray are tested by idle
        // lanes in this same iteration — lane k (or k + 32) takes triangle t_base + k with the owner's ray — and folded back into the owner.
        bool helping = false; int owner = -1; uint32_t help_pk = 0;
        if (draining) {
            const bool coop_kind = (tagw >> 31) != 0 ? (MRT_COOP_MODE & 1) != 0 : (MRT_COOP_MODE & 2) != 0;
            const unsigned long long m_many = __ballot(live && coop_kind && (t_mask & (t_mask - 1u)) != 0u && (!TWO_LEVEL || in_blas));
            const unsigned long long m_free = __ballot(!live && !unreported);
            if (m_many != 0ull && m_free != 0ull) {
                owner = __ffsll((long long)m_many) - 1;
                const uint32_t o_tb = (uint32_t)__builtin_amdgcn_readlane((int)t_base, owner), o_tm = (uint32_t)__builtin_amdgcn_readlane((int)t_mask, owner);
                const uint32_t f_lo = (uint32_t)m_free, f_hi = (uint32_t)(m_free >> 32), bit = lane & 31u;
                helping = !live && !unreported && ((o_tm >> bit) & 1u) != 0u && (lane < 32u || ((f_lo >> bit) & 1u) == 0u);
                // the owner's ray is read HERE, where the whole wave is on: the owner is not among the helpers, and inside `if (helping)` a register the compiler had to reload would
                // hold the owner's lane no more (a reload covers the active lanes only — two variant builds with more scratch rendered wrong images, tests/test_kernel_resources.py)
                const f3 o_o = mk3(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(o.x), owner)), __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(o.y), owner)),
                                   __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(o.z), owner)));
                const f3 o_d = mk3(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d.x), owner)), __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d.y), owner)),
                                   __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d.z), owner)));
                const float o_bt = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(best_t), owner));
                if (helping) { o = o_o; d = o_d; best_t = o_bt; help_pk = o_tb + bit; }
                if ((int)lane == owner) t_mask &= ~((f_lo | f_hi) & o_tm);      // these are being tested now
            }
        }
        if constexpr (Pairs::on) {
            // TLAS pass: every LARGE instance among the lane's pending ones goes to the pair queue now (a short divergent loop: at most eight trips); the small ones stay pending and are
            // entered in place, one per iteration, as without a queue.  A lane whose pending instances were all large goes on to its next node in this same iteration.
            if (live && t_mask != 0u && !in_blas && !pq_refused) {
                uint32_t rest = t_mask, keep = 0u;
                while (rest != 0u) {
                    const uint32_t k = (uint32_t)__ffs((int)rest) - 1u; rest &= rest - 1u;
                    uint32_t tl_slot = t_base + k;
                    MRT_BOUND(tl_slot, s.num_wtlas, 3);
                    uint32_t id = s.wtlas_index[tl_slot];
                    MRT_BOUND(id, s.num_inst, 4);
                    if (s.inst[id].ntri > 8u && !pq_refused) {
                        const bool pushed = pq.push(qi, id, best_t, tagw);
                        if (!pushed) { pq_refused = true; keep |= 1u << k; }      // refused (the queue is full): this lane walks its large instances in place from here on
                    }
                    else keep |= 1u << k;
                }
                t_mask = keep;
            }
        }
        const bool has_inst = TWO_LEVEL && live && t_mask != 0 && !in_blas;      // at the TLAS level a pending "triangle" is an instance to enter
        bool has_tri = live && t_mask != 0 && !has_inst;
        uint32_t t_rest = t_mask & (t_mask - 1u);               // triangles left after this iteration's first one
        bool want_node = live && (t_rest == 0u || (SPEC && u_mask == 0u && (!TWO_LEVEL || in_blas))) && !has_inst;      // a place for the node's triangles after this iteration's test
        uint32_t pending = 0, tri_pk = 0;
        bool last_step = false;         // flattened scenes: nothing but this iteration's triangle is left of the ray — it is finished when the test is done
        if (TWO_LEVEL && has_inst) {
            // enter the next instance of the TLAS leaf: park the TLAS group (always, also without siblings left: the exit pops it), take the ray
            // into object space, and fetch the BLAS root in this same iteration
            const uint32_t k = (uint32_t)__ffs((int)t_mask) - 1u;
            t_mask &= t_mask - 1u;
            uint32_t tl_slot = t_base + k;
            MRT_BOUND(tl_slot, s.num_wtlas, 3);
            uint32_t id = s.wtlas_index[tl_slot];
            MRT_BOUND(id, s.num_inst, 4);
            const InstanceDev &I = s.inst[id];
            tl_pack = (t_base << 8) | t_mask;
            uint32_t sp = (g_mask >> 16) & 0xFFu;
            wstack_push(stack, sp, lane, g_base, g_mask & 0xFFFFu); sp++;
            const f3 oo = to_object_point(I, o), dd = to_object_dir(I, d);
            o = oo; d = dd; ix = box_inv(d.x); iy = box_inv(d.y); iz = box_inv(d.z);
            nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f; oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
            g_base = 0; g_mask = (sp << 24) | (sp << 16);          // no group yet: the root's children become the first one
            t_base = 0; t_mask = 0;
            insts = (insts & 0xFFFF0000u) | id;
            if (I.ntri <= 8u) {          // a BLAS of a few triangles (a wall, a floor): no node to test, its packets are the pending set, the first one is tested now
                t_base = I.packet_base; t_mask = (1u << I.ntri) - 1u; t_rest = t_mask & (t_mask - 1u); has_tri = true;
            } else { pending = I.wroot; want_node = true; }
        }
        else if (want_node) {
            if ((g_mask & 0xFF00u) == 0) {
                const uint32_t sp = TWO_LEVEL ? (g_mask >> 16) & 0xFFu : g_mask >> 16, isp = TWO_LEVEL ? g_mask >> 24 : 0u;
                if (TWO_LEVEL && isp != 0u && sp == isp) want_node = false;      // the BLAS's last triangle is tested in this iteration; the lane leaves in the next
                else if (sp == 0) {         // no node left: the ray ends with its last triangle (helpers may just have taken the first group's: the second one still counts)
                    want_node = false;
                    if (!has_tri) { if (!SPEC || u_mask == 0u) { live = false; unreported = true; } }
                    else if (!TWO_LEVEL && t_rest == 0u && u_mask == 0u) last_step = true;
                }
                else { wstack_pop(stack, sp - 1u, lane, g_base, g_mask); g_mask |= ((sp - 1u) << 16) | (isp << 24); }
            }
            if (want_node) {
                const uint32_t hits = (g_mask >> 8) & 0xFFu;
                const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;       // nearest remaining child in (slot ^ octant) order
                g_mask &= ~(0x100u << b);
                const uint32_t slot = b ^ oct;
                pending = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));
            }
        }
        if (ss) { ss->tri_sum += (uint32_t)__popcll(__ballot(has_tri)); ss->node_sum += (uint32_t)__popcll(__ballot(want_node)); }
#if defined(MRT_STATS_PROBE) && MRT_STATS_PROBE == 1
        if (ss) for (int k = 0; k < 3; k++) ss->probe[k] += (uint32_t)__popcll(__ballot(want_node && pending < ss->level_end[k]));
#elif defined(MRT_STATS_PROBE) && MRT_STATS_PROBE == 2
        if (ss) {
            const uint32_t pend = live ? (uint32_t)__popc(t_mask) + (SPEC ? (uint32_t)__popc(u_mask) : 0u) : 0u;
            uint32_t tot = pend;
            for (int o_ = 32; o_ > 0; o_ >>= 1) tot += (uint32_t)__shfl_xor((int)tot, o_);
            ss->probe[0] += tot; ss->probe[1] += (uint32_t)__popcll(__ballot(pend >= 2u)); ss->probe[2] += __popcll(__ballot(has_tri)) < 16 ? 1u : 0u;
        }
#endif
#ifdef MRT_STATS_BOTH      // diagnostics build (tools/archive/two_level_probe.py): in place of the refill counters, lane-iterations that do a triangle AND a node / that enter an instance
        if (ss) { ss->refills += (uint32_t)__popcll(__ballot(has_tri && want_node)); ss->refill_lanes += (uint32_t)__popcll(__ballot(TWO_LEVEL && has_inst)); }
#endif
        float4 r0, r1, r2, n0, n1, n2, n3, n4;        // loaded under has_tri / want_node and used under the same predicates;
        // an empty asm "defines" them on the other paths without the 28 v_mov a zero initialiser costs per iteration
        asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r2.x), "=v"(r2.y), "=v"(r2.z));
        asm volatile("" : "=v"(n0.x), "=v"(n0.y), "=v"(n0.z), "=v"(n0.w), "=v"(n1.x), "=v"(n1.y), "=v"(n1.z), "=v"(n1.w), "=v"(n2.x), "=v"(n2.y), "=v"(n2.z), "=v"(n2.w));
        asm volatile("" : "=v"(n3.x), "=v"(n3.y), "=v"(n3.z), "=v"(n3.w), "=v"(n4.x), "=v"(n4.y), "=v"(n4.z), "=v"(n4.w));
        r1.w = 0.0f; r2.w = 0.0f;
        if (has_tri || helping) {
            if (has_tri) { tri_pk = t_base + (uint32_t)__ffs((int)t_mask) - 1u; t_mask = t_rest; }
            else tri_pk = help_pk;
            MRT_BOUND(tri_pk, s.num_wpackets, 2);
            const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)tri_pk;
            r0 = pk[0]; r1 = pk[1]; r2 = pk[2];
        }
        if (want_node) {
            MRT_BOUND(pending, s.num_wnodes, 1);
            const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending;
            n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4];
        }
        auto consider = [&](const float4 q0, const float4 q1, const float4 q2, const uint32_t pk_index) {
            float t, U, V, ad;
            if (tri_test(q0, q1, q2, o, d, 0.0f, best_t, t, U, V, ad)) {
                if ((tagw >> 31) != 0) { best_pk = pk_index; live = false; unreported = true; }   // any-hit ray: done
                else {
                    bool better = t < best_t || best_pk == 0xFFFFFFFFu;
                    if (!better) {                                      // t == best_t: ties go to the lowest (global) id (rare)
                        if (TWO_LEVEL) better = s.inst[insts & 0xFFFFu].gid_base + __float_as_uint(q0.w) < s.inst[insts >> 16].gid_base + __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);
                        else if (Ext::hit_lds) better = __float_as_uint(q0.w) < __float_as_uint(ext.hit[192u + lane]);
                        else better = __float_as_uint(q0.w) < __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);
                    }
                    if (better) {
                        best_t = t; best_pk = pk_index; if (TWO_LEVEL) insts = (insts & 0xFFFFu) | (insts << 16);
                        if (Ext::hit_lds) { ext.hit[lane] = U; ext.hit[64u + lane] = V; ext.hit[128u + lane] = ad; ext.hit[192u + lane] = q0.w; }
                    }
                }
            }
        };
        if (has_tri) consider(r0, r1, r2, tri_pk);
        if (owner >= 0) {                                        // wave-uniform
            float h_t = 0.0f; bool h_hit = false;
            float U_ = 0.0f, V_ = 0.0f, ad_ = 1.0f;
            if (helping) h_hit = tri_test(r0, r1, r2, o, d, 0.0f, best_t, h_t, U_, V_, ad_);
            // the closest of the helpers' hits (ties: lowest id — all of them are triangles of the owner's current BLAS, so local ids compare)
            float bt = __builtin_inff(); uint32_t bpk = 0xFFFFFFFFu, bgid = 0xFFFFFFFFu; int bl = 0;
            for (unsigned long long m = __ballot(h_hit); m != 0ull; m &= m - 1ull) {
                const int l = __ffsll((long long)m) - 1;
                const float t_ = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(h_t), l));
                const uint32_t g_ = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(r0.w), l), p_ = (uint32_t)__builtin_amdgcn_readlane((int)tri_pk, l);
                if (t_ < bt || (t_ == bt && g_ < bgid)) { bt = t_; bgid = g_; bpk = p_; bl = l; }
            }
            // hit_lds: the winning helper's U, V, |det| travel to the owner with its distance
            const float wU = Ext::hit_lds ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(U_), bl)) : 0.0f, wV = Ext::hit_lds ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(V_), bl)) : 0.0f,
                        wA = Ext::hit_lds ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ad_), bl)) : 1.0f;
            // (the owner may have run out of nodes in this very iteration — it then waits for its report with these hits folded in)
            if ((int)lane == owner && bpk != 0xFFFFFFFFu) {
                if ((tagw >> 31) != 0) { best_pk = bpk; live = false; unreported = true; }
                else {
                    bool better = bt < best_t || best_pk == 0xFFFFFFFFu;
                    if (!better && bt == best_t) {
                        if (TWO_LEVEL) better = s.inst[insts & 0xFFFFu].gid_base + bgid < s.inst[insts >> 16].gid_base + __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);
                        else if (Ext::hit_lds) better = bgid < __float_as_uint(ext.hit[192u + lane]);
                        else better = bgid < __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);
                    }
                    if (better) {
                        best_t = bt; best_pk = bpk; if (TWO_LEVEL) insts = (insts & 0xFFFFu) | (insts << 16);
                        if (Ext::hit_lds) { ext.hit[lane] = wU; ext.hit[64u + lane] = wV; ext.hit[128u + lane] = wA; ext.hit[192u + lane] = __uint_as_float(bgid); }
                    }
                }
            }
        }
        if (want_node && live) {
            uint32_t node_hits, tri_hits;
            wide_node_test<MRT_WIDE_SCALED != 0>(n0, n1, n2, n3, n4, o, ix, iy, iz, nx, ny, nz, oct, 0.0f, best_t, node_hits, tri_hits);
            uint32_t sp = TWO_LEVEL ? (g_mask >> 16) & 0xFFu : g_mask >> 16;
            const uint32_t isp = TWO_LEVEL ? g_mask & 0xFF000000u : 0u;
            if ((g_mask & 0xFF00u) != 0) { wstack_push(stack, sp, lane, g_base, g_mask & 0xFFFFu); sp++; }     // siblings still to visit
            g_base = __float_as_uint(n1.x); g_mask = isp | (sp << 16) | (node_hits << 8) | (__float_as_uint(n0.w) >> 24);
            if (SPEC && t_mask != 0u) { u_base = __float_as_uint(n1.y); u_mask = tri_hits; }      // (u is empty here: the lane asked for a node with t_rest != 0 only then)
            else { t_base = __float_as_uint(n1.y); t_mask = tri_hits; }
        }
        if (last_step && live) { live = false; unreported = true; }
#ifdef MRT_PROBE_EXTRA_VALU      // diagnostics build: N more full-rate VALU instructions per iteration, on a scratch register (how much of the loop's time is VALU issue?)
        { float sink_; asm volatile("v_mov_b32 %0, %1" : "=v"(sink_) : "v"(best_t));
#pragma unroll
          for (int k_ = 0; k_ < MRT_PROBE_EXTRA_VALU; k_++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(sink_)); }
#endif
    }
}

}  // namespace
}  // namespace mrt
