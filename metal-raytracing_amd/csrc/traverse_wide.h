// traverse_wide.h — traversal of the 8-wide compressed layout (scene_device.h "wide node") with a short
// per-lane stack held in LDS.  Second backend for the two uses of Apple's opaque
// `intersector.intersect` (Raytracing.metal:244 closest, :367 any).
//
// Why it exists (measured on MI355X): with the binary rope layout every box test costs one
// dependent memory round trip of 3 x 16 B per lane, and the traversal kernels were bound by the vector
// memory path (TA/L1 request rate on divergent 16-B gathers) and by the length of the dependent chain of
// the slowest ray — not by arithmetic.  One 80-byte wide node tests eight quantised child boxes per
// round trip: ~4x fewer 16-B requests per box and ~3x fewer round trips per ray.  The stack holds
// (child_base, hit mask | imask) pairs, 8 B per entry, at most one entry per tree level; it lives in LDS
// ([entry][lane], conflict free), never in scratch or HBM.
//
// The result is identical to the rope backend bit for bit: the closest hit is the global minimum t
// with ties to the lowest triangle id, and the quantised boxes only ever grow.
#pragma once
#include "traverse.h"

namespace mrt {
namespace {

typedef float float2v __attribute__((ext_vector_type(2)));
#ifndef MRT_WIDE_PK_FMA
#define MRT_WIDE_PK_FMA 0   // measured: the operand pairs cost ~3 VGPRs -> spills at the 80-register budget; -2.6 %
#endif
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

// stack: depth x WIDE_STACK_LEVEL_BYTES of LDS for the wave (depth = the scene's wide-tree depth, <= WIDE_STACK)
template <bool ANY, bool STATS = false, bool RUNTIME_ANY = false>
MRT_DEV bool traverse_wide(const SceneView &s, f3 o, f3 d, float tmin, float tmax, TravHit &h, uint32_t *stack /* depth x WIDE_STACK_LEVEL_BYTES in LDS */, TravCounters *tc = nullptr, bool any_rt = false) {
    h.t = tmax; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu;
    if (s.num_wnodes == 0) return false;
    const uint32_t lane = threadIdx.x & 63;
    const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);
    const bool nx = d.x < 0.0f, ny = d.y < 0.0f, nz = d.z < 0.0f;
    const uint32_t oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
    uint32_t sp = 0;
    uint32_t g_base = 0, g_mask = 0;      // node group: (permuted hit bits << 8) | imask
    uint32_t t_base = 0, t_mask = 0;      // triangle group: bit k = packet t_base + k still to test
    uint32_t pending = 0; bool have_pending = true;     // start by entering the root
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
            const float4 *__restrict__ pk = s.wpackets + 3 * (size_t)(t_base + k);
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
            const uint32_t ew = __float_as_uint(n0.w);
            const uint32_t imask = ew >> 24;
            // t = q * (2^e * idir) + (p - o) * idir per plane
            const float ax = __uint_as_float((ew & 0xFFu) << 23) * ix, ay = __uint_as_float(((ew >> 8) & 0xFFu) << 23) * iy, az = __uint_as_float(((ew >> 16) & 0xFFu) << 23) * iz;
            const float bx = (n0.x - o.x) * ix, by = (n0.y - o.y) * iy, bz = (n0.z - o.z) * iz;
            // near / far plane bytes per axis: swap lo and hi where the direction is negative
            const uint32_t lx0 = __float_as_uint(n2.x), lx1 = __float_as_uint(n2.y), ly0 = __float_as_uint(n2.z), ly1 = __float_as_uint(n2.w);
            const uint32_t lz0 = __float_as_uint(n3.x), lz1 = __float_as_uint(n3.y), hx0 = __float_as_uint(n3.z), hx1 = __float_as_uint(n3.w);
            const uint32_t hy0 = __float_as_uint(n4.x), hy1 = __float_as_uint(n4.y), hz0 = __float_as_uint(n4.z), hz1 = __float_as_uint(n4.w);
            const uint32_t nrx[2] = {nx ? hx0 : lx0, nx ? hx1 : lx1}, frx[2] = {nx ? lx0 : hx0, nx ? lx1 : hx1};
            const uint32_t nry[2] = {ny ? hy0 : ly0, ny ? hy1 : ly1}, fry[2] = {ny ? ly0 : hy0, ny ? ly1 : hy1};
            const uint32_t nrz[2] = {nz ? hz0 : lz0, nz ? hz1 : lz1}, frz[2] = {nz ? lz0 : hz0, nz ? lz1 : hz1};
            const uint32_t meta[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)};
            uint32_t node_hits = 0, tri_hits = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int w = i >> 2, k = i & 3;
                const float tn = fmaxf(fmaxf(__builtin_fmaf(ubyte_f(nrx[w], k), ax, bx), __builtin_fmaf(ubyte_f(nry[w], k), ay, by)),
                                       fmaxf(__builtin_fmaf(ubyte_f(nrz[w], k), az, bz), tmin));
                const float tf = fminf(fminf(fminf(__builtin_fmaf(ubyte_f(frx[w], k), ax, bx), __builtin_fmaf(ubyte_f(fry[w], k), ay, by)),
                                             __builtin_fmaf(ubyte_f(frz[w], k), az, bz)) * 1.0000005f, h.t);
                // decode error of the fused plane evaluation is far below the build-time padding of the leaf boxes
                if (tn <= tf) {
                    if ((imask >> i) & 1u) node_hits |= 1u << ((uint32_t)i ^ oct);
                    else { const uint32_t m = (meta[w] >> (8 * k)) & 0xFFu; tri_hits |= ((1u << (m >> 5)) - 1u) << (m & 31u); }
                }
            }
            if (STATS && tri_hits) tc->leaves++;
            if ((g_mask >> 8) != 0) { wstack_push(stack, sp, lane, g_base, g_mask); sp++; }     // siblings still to visit
            g_base = __float_as_uint(n1.x); g_mask = (node_hits << 8) | imask;
            t_base = __float_as_uint(n1.y); t_mask = tri_hits;
        }
    }
    return h.gid != 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------
// Wave-level stream traversal with lane refill (the "persistent wavefront + ray compaction" of BASELINE.json
// configs[4]) on the wide layout.  rocprofv3 on the one-ray-per-lane kernel: VALUBusy 81 %, VALUUtilization 19 % —
// the VALU is saturated by instructions in which four lanes out of five are idle, because the rays of a wave
// finish at very different times.  Here a wave owns `per_wave` consecutive rays of the queue; the next 64 of
// them are PREFETCHED into registers (origin, direction, 1/direction — computed by all 64 lanes at full
// occupancy), and whenever REFILL_AT lanes are idle they take the next prefetched rays by cross-lane reads
// (ds_bpermute), without touching memory.  A refilled lane starts with an empty stack, so nothing but the ray
// moves.  `emit(idx, B.w, hit, h)` is called by a lane when its ray finishes.
#ifndef MRT_WIDE_REFILL_AT
#define MRT_WIDE_REFILL_AT 16
#endif
constexpr int WIDE_REFILL_AT = MRT_WIDE_REFILL_AT;

struct StreamStats { uint32_t iters, live_sum, node_sum, tri_sum, refills, refill_lanes; };

// Batch sources for traverse_wide_stream: `next(base, limit)` hands the wave its next (up to) 64 rays [base, min(base+64, limit)).
struct StaticBatches {          // a fixed range of consecutive rays per wave
    uint32_t cur, end;
    MRT_DEV bool next(uint32_t &base, uint32_t &limit) { if (cur >= end) return false; base = cur; limit = end; cur += 64; return true; }
};
struct SharedBatches {          // the waves of one shard pull 64-ray batches from a shared counter: no wave idles while its shard has rays left
    uint32_t *counter; uint32_t begin, end;
    MRT_DEV bool next(uint32_t &base, uint32_t &limit) {
        uint32_t b = 0;
        if ((threadIdx.x & 63) == 0) b = atomicAdd(counter, 1u);
        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        const unsigned long long off = (unsigned long long)b * 64ull;
        if (off >= (unsigned long long)(end - begin)) return false;
        base = begin + (uint32_t)off; limit = end;
        return true;
    }
};

template <class BatchSrc, class RayFetch, class Emit>
MRT_DEV void traverse_wide_stream(const SceneView &s, BatchSrc src_batches, uint32_t *stack, RayFetch fetch, Emit emit,
                                  const float4 *lds_top = nullptr, uint32_t n_top = 0, StreamStats *ss = nullptr) {
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // prefetched batch: ray (batch_base + lane)
    float4 pA = make_float4(0, 0, 0, 0), pB = pA;     // pB.w = pixel | any-hit flag << 31 (register budget: 64 VGPRs = 8 waves per SIMD)
    uint32_t batch_base = 0, batch_end = 0, batch_used = 64;      // wave-uniform; used == 64 -> nothing prefetched
    bool more_batches = true;
    // live ray
    bool live = false, unreported = false;            // unreported: the lane's ray is finished, its result not yet emitted
    uint32_t idx = 0, pixw = 0;                       // pixw bit 31 = any-hit ray
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1); float ix = 0, iy = 0, iz = 0; bool nx = false, ny = false, nz = false; uint32_t oct = 0;
    TravHit h; h.t = 0; h.U = 0; h.V = 0; h.ad = 1; h.gid = 0xFFFFFFFFu;
    uint32_t sp = 0, g_base = 0, g_mask = 0, t_base = 0, t_mask = 0, best_pk = 0;
    for (;;) {
        const unsigned long long m_idle = __ballot(!live);
        const uint32_t n_idle = (uint32_t)__popcll(m_idle);
        bool refilled = false;
        if (n_idle >= (uint32_t)WIDE_REFILL_AT || m_idle == ~0ull) {
            // results are written here, by all idle lanes together, not in the iteration a ray happens to finish in
            // (the divisions and stores of `emit` would otherwise run at one or two lanes per iteration)
            if (unreported) {
                const bool was_any = (pixw >> 31) != 0, was_hit = h.gid != 0xFFFFFFFFu;
                if (!was_any && was_hit) {      // barycentrics of the winning triangle: recomputed here (same arithmetic) instead of living in 3 registers
                    const float4 *__restrict__ pk = s.wpackets + 3 * (size_t)best_pk;
                    float t_;
                    (void)tri_test(pk[0], pk[1], pk[2], o, d, 0.0f, __builtin_inff(), t_, h.U, h.V, h.ad);
                }
                emit(idx, pixw & 0x7FFFFFFFu, was_any, was_hit, h); unreported = false;
            }
            if (batch_used >= 64 && more_batches) {            // prefetch the next 64 rays (coalesced), all lanes
                more_batches = src_batches.next(batch_base, batch_end);
                if (more_batches) batch_used = 0;
                const uint32_t i = batch_base + lane;
                if (more_batches && i < batch_end) {
                    uint32_t p_any = 0;
                    fetch(i, pA, pB, p_any);
                    pB.w = __uint_as_float((__float_as_uint(pB.w) & 0x7FFFFFFFu) | (p_any << 31));
                }
            }
            const uint32_t avail = batch_used < 64 ? min(64u - batch_used, batch_end > batch_base + batch_used ? batch_end - (batch_base + batch_used) : 0u) : 0u;
            if (avail == 0) { if (m_idle == ~0ull) break; }
            else {
                const uint32_t rank = (uint32_t)__popcll(m_idle & lt);
                const uint32_t src = batch_used + rank;                    // lane holding my new ray
                const bool take = !live && rank < avail;
                // cross-lane reads are executed by every lane (bpermute needs the whole wave)
                const int sl = (int)(take ? src : lane);
                const float ax_ = __shfl(pA.x, sl), ay_ = __shfl(pA.y, sl), az_ = __shfl(pA.z, sl), aw_ = __shfl(pA.w, sl);
                const float bx_ = __shfl(pB.x, sl), by_ = __shfl(pB.y, sl), bz_ = __shfl(pB.z, sl), bw_ = __shfl(pB.w, sl);
                if (take) {
                    o = mk3(ax_, ay_, az_); d = mk3(bx_, by_, bz_); ix = safe_inv(bx_); iy = safe_inv(by_); iz = safe_inv(bz_);
                    nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f; oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                    h.t = aw_; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu;
                    idx = batch_base + src; pixw = __float_as_uint(bw_);
                    // enter the root as the only "hit child" of a pseudo group: base 0, no internal-child bits -> node 0
                    sp = 0; g_base = 0; g_mask = s.num_wnodes != 0 ? 0x100u : 0u; t_base = 0; t_mask = 0;
                    live = true;
                }
                batch_used += min(avail, n_idle);
                refilled = true;
                if (ss) { ss->refills++; ss->refill_lanes += min(avail, n_idle); }
            }
        }
        if (refilled) continue;
        if (ss) { ss->iters++; ss->live_sum += (uint32_t)__popcll(__ballot(live)); ss->tri_sum += (uint32_t)__popcll(__ballot(live && t_mask != 0)); ss->node_sum += (uint32_t)__popcll(__ballot(live && t_mask == 0)); }
#ifndef MRT_WIDE_PIPE
#define MRT_WIDE_PIPE 1
#endif
#if MRT_WIDE_PIPE
        // One memory round trip per iteration.  A lane with at most one triangle left to test already knows the next node
        // it will visit (the nearest remaining hit child, or the top of its stack), so it fetches that node (80 B) together
        // with the triangle packet (48 B), tests the triangle, then the node's eight boxes against the possibly shorter ray.
        const bool has_tri = live && t_mask != 0;
        bool want_node = live && (t_mask & (t_mask - 1u)) == 0u;
        uint32_t pending = 0, tri_pk = 0;
        if (want_node) {
            if ((g_mask >> 8) == 0) {
                if (sp == 0) { want_node = false; if (!has_tri) { live = false; unreported = true; } }
                else { sp--; wstack_pop(stack, sp, lane, g_base, g_mask); }
            }
            if (want_node) {
                const uint32_t hits = g_mask >> 8;
                const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;
                g_mask &= ~(0x100u << b);
                const uint32_t slot = b ^ oct;
                pending = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));
            }
        }
#if MRT_WIDE_PIPE == 2      // unpredicated loads: lanes without that kind of work read element 0
        float4 r0, r1, r2, n0, n1, n2, n3, n4;
        {
            uint32_t k = 0;
            if (has_tri) { k = (uint32_t)__ffs((int)t_mask) - 1u; t_mask &= t_mask - 1u; tri_pk = t_base + k; }
            const float4 *__restrict__ pk = s.wpackets + 3 * (size_t)tri_pk;
            const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending;
            r0 = pk[0]; r1 = pk[1]; r2 = pk[2];
            n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4];
        }
#else
        float4 r0, r1, r2, n0, n1, n2, n3, n4;        // loaded under has_tri / want_node and used under the same predicates;
        // an empty asm "defines" them on the other paths without the 28 v_mov a zero initialiser costs per iteration
        asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r2.x), "=v"(r2.y), "=v"(r2.z));
        asm volatile("" : "=v"(n0.x), "=v"(n0.y), "=v"(n0.z), "=v"(n0.w), "=v"(n1.x), "=v"(n1.y), "=v"(n1.z), "=v"(n1.w), "=v"(n2.x), "=v"(n2.y), "=v"(n2.z), "=v"(n2.w));
        asm volatile("" : "=v"(n3.x), "=v"(n3.y), "=v"(n3.z), "=v"(n3.w), "=v"(n4.x), "=v"(n4.y), "=v"(n4.z), "=v"(n4.w));
        r1.w = 0.0f; r2.w = 0.0f;
        if (has_tri) {
            const uint32_t k = (uint32_t)__ffs((int)t_mask) - 1u;
            t_mask &= t_mask - 1u;
            tri_pk = t_base + k;
            const float4 *__restrict__ pk = s.wpackets + 3 * (size_t)tri_pk;
            r0 = pk[0]; r1 = pk[1]; r2 = pk[2];
        }
        if (want_node) {
            if (pending < n_top) { const float4 *__restrict__ nd = lds_top + 5u * pending; n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4]; }
            else { const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending; n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4]; }
        }
#endif
        if (has_tri) {
            float t, U, V, ad;
            if (tri_test(r0, r1, r2, o, d, 0.0f, h.t, t, U, V, ad)) {
                const uint32_t gid = __float_as_uint(r0.w);
                if ((pixw >> 31) != 0) { h.gid = gid; live = false; unreported = true; }
                else if (t < h.t || gid < h.gid) { h.t = t; h.gid = gid; best_pk = tri_pk; }
            }
        }
        if (want_node && live) {
            const uint32_t ew = __float_as_uint(n0.w);
            const uint32_t imask = ew >> 24;
            const float ax = __uint_as_float((ew & 0xFFu) << 23) * ix, ay = __uint_as_float(((ew >> 8) & 0xFFu) << 23) * iy, az = __uint_as_float(((ew >> 16) & 0xFFu) << 23) * iz;
            const float bx = (n0.x - o.x) * ix, by = (n0.y - o.y) * iy, bz = (n0.z - o.z) * iz;
            const uint32_t lx0 = __float_as_uint(n2.x), lx1 = __float_as_uint(n2.y), ly0 = __float_as_uint(n2.z), ly1 = __float_as_uint(n2.w);
            const uint32_t lz0 = __float_as_uint(n3.x), lz1 = __float_as_uint(n3.y), hx0 = __float_as_uint(n3.z), hx1 = __float_as_uint(n3.w);
            const uint32_t hy0 = __float_as_uint(n4.x), hy1 = __float_as_uint(n4.y), hz0 = __float_as_uint(n4.z), hz1 = __float_as_uint(n4.w);
            const uint32_t nrx[2] = {nx ? hx0 : lx0, nx ? hx1 : lx1}, frx[2] = {nx ? lx0 : hx0, nx ? lx1 : hx1};
            const uint32_t nry[2] = {ny ? hy0 : ly0, ny ? hy1 : ly1}, fry[2] = {ny ? ly0 : hy0, ny ? ly1 : hy1};
            const uint32_t nrz[2] = {nz ? hz0 : lz0, nz ? hz1 : lz1}, frz[2] = {nz ? lz0 : hz0, nz ? lz1 : hz1};
            const uint32_t meta[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)};
            uint32_t node_hits = 0, tri_hits = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int w = i >> 2, k = i & 3;
#if MRT_WIDE_PK_FMA        // near and far plane of an axis in one v_pk_fma_f32 (packed fp32 runs at twice the scalar rate)
                const float2v px = __builtin_elementwise_fma(float2v{ubyte_f(nrx[w], k), ubyte_f(frx[w], k)}, float2v{ax, ax}, float2v{bx, bx});
                const float2v py = __builtin_elementwise_fma(float2v{ubyte_f(nry[w], k), ubyte_f(fry[w], k)}, float2v{ay, ay}, float2v{by, by});
                const float2v pz = __builtin_elementwise_fma(float2v{ubyte_f(nrz[w], k), ubyte_f(frz[w], k)}, float2v{az, az}, float2v{bz, bz});
                const float tn = fmaxf(fmaxf(px.x, py.x), fmaxf(pz.x, 0.0f));
                const float tf = fminf(fminf(fminf(px.y, py.y), pz.y) * 1.0000005f, h.t);
#else
                const float tn = fmaxf(fmaxf(__builtin_fmaf(ubyte_f(nrx[w], k), ax, bx), __builtin_fmaf(ubyte_f(nry[w], k), ay, by)),
                                       fmaxf(__builtin_fmaf(ubyte_f(nrz[w], k), az, bz), 0.0f));
                const float tf = fminf(fminf(fminf(__builtin_fmaf(ubyte_f(frx[w], k), ax, bx), __builtin_fmaf(ubyte_f(fry[w], k), ay, by)),
                                             __builtin_fmaf(ubyte_f(frz[w], k), az, bz)) * 1.0000005f, h.t);
#endif
                if (tn <= tf) {
                    if ((imask >> i) & 1u) node_hits |= 1u << ((uint32_t)i ^ oct);
                    else { const uint32_t m = (meta[w] >> (8 * k)) & 0xFFu; tri_hits |= ((1u << (m >> 5)) - 1u) << (m & 31u); }
                }
            }
#ifdef MRT_PROBE_EXTRA_LOADS      // bottleneck probe: extra divergent 16-B loads per node visit (result folded into a never-true test)
            for (int r = 0; r < MRT_PROBE_EXTRA_LOADS; r++) {
                const float4 x = s.wnodes[WNODE_STRIDE * (size_t)((pending * 2654435761u + 977u * (r + 1)) % s.num_wnodes) + (r % 5)];
                if (x.x == 1.2345e-30f) tri_hits |= 1u;
            }
#endif
#ifdef MRT_PROBE_EXTRA_VALU       // bottleneck probe: extra dependent VALU work per node visit
            { float acc = bx; for (int r = 0; r < MRT_PROBE_EXTRA_VALU; r++) acc = __builtin_fmaf(acc, ax, by); if (acc == 1.2345e-30f) tri_hits |= 1u; }
#endif
            if ((g_mask >> 8) != 0) { wstack_push(stack, sp, lane, g_base, g_mask); sp++; }
            g_base = __float_as_uint(n1.x); g_mask = (node_hits << 8) | imask;
            t_base = __float_as_uint(n1.y); t_mask = tri_hits;
        }
#else
#ifndef MRT_WIDE_ORDER
#define MRT_WIDE_ORDER 1
#endif
        // One iteration = node phase, then triangle phase.  A lane whose node visit produced leaf triangles tests the first
        // of them in the same iteration (MRT_WIDE_ORDER 1): the wave pays for both phases whenever its lanes disagree anyway.
        const bool was_tri = t_mask != 0;
        if (live && !was_tri) {
            bool finished = false;
            if ((g_mask >> 8) == 0) {
                if (sp == 0) finished = true;
                else { sp--; wstack_pop(stack, sp, lane, g_base, g_mask); }
            }
            if (finished) { live = false; unreported = true; }
            else {
                const uint32_t hits = g_mask >> 8;
                const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;
                g_mask &= ~(0x100u << b);
                const uint32_t slot = b ^ oct;
                const uint32_t pending = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));     // the node visited now
                float4 n0, n1, n2, n3, n4;
                if (pending < n_top) {          // the top levels of the wide tree (BFS numbering) are staged in LDS by the workgroup
                    const float4 *__restrict__ nd = lds_top + 5u * pending;
                    n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4];
                } else {
                    const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending;
                    n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4];
                }
                const uint32_t ew = __float_as_uint(n0.w);
                const uint32_t imask = ew >> 24;
                const float ax = __uint_as_float((ew & 0xFFu) << 23) * ix, ay = __uint_as_float(((ew >> 8) & 0xFFu) << 23) * iy, az = __uint_as_float(((ew >> 16) & 0xFFu) << 23) * iz;
                const float bx = (n0.x - o.x) * ix, by = (n0.y - o.y) * iy, bz = (n0.z - o.z) * iz;
                const uint32_t lx0 = __float_as_uint(n2.x), lx1 = __float_as_uint(n2.y), ly0 = __float_as_uint(n2.z), ly1 = __float_as_uint(n2.w);
                const uint32_t lz0 = __float_as_uint(n3.x), lz1 = __float_as_uint(n3.y), hx0 = __float_as_uint(n3.z), hx1 = __float_as_uint(n3.w);
                const uint32_t hy0 = __float_as_uint(n4.x), hy1 = __float_as_uint(n4.y), hz0 = __float_as_uint(n4.z), hz1 = __float_as_uint(n4.w);
                const uint32_t nrx[2] = {nx ? hx0 : lx0, nx ? hx1 : lx1}, frx[2] = {nx ? lx0 : hx0, nx ? lx1 : hx1};
                const uint32_t nry[2] = {ny ? hy0 : ly0, ny ? hy1 : ly1}, fry[2] = {ny ? ly0 : hy0, ny ? ly1 : hy1};
                const uint32_t nrz[2] = {nz ? hz0 : lz0, nz ? hz1 : lz1}, frz[2] = {nz ? lz0 : hz0, nz ? lz1 : hz1};
                const uint32_t meta[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)};
                uint32_t node_hits = 0, tri_hits = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int w = i >> 2, k = i & 3;
                    const float tn = fmaxf(fmaxf(__builtin_fmaf(ubyte_f(nrx[w], k), ax, bx), __builtin_fmaf(ubyte_f(nry[w], k), ay, by)),
                                           fmaxf(__builtin_fmaf(ubyte_f(nrz[w], k), az, bz), 0.0f));
                    const float tf = fminf(fminf(fminf(__builtin_fmaf(ubyte_f(frx[w], k), ax, bx), __builtin_fmaf(ubyte_f(fry[w], k), ay, by)),
                                                 __builtin_fmaf(ubyte_f(frz[w], k), az, bz)) * 1.0000005f, h.t);
                    if (tn <= tf) {
                        if ((imask >> i) & 1u) node_hits |= 1u << ((uint32_t)i ^ oct);
                        else { const uint32_t m = (meta[w] >> (8 * k)) & 0xFFu; tri_hits |= ((1u << (m >> 5)) - 1u) << (m & 31u); }
                    }
                }
#ifdef MRT_PROBE_EXTRA_LOADS      // bottleneck probe: extra divergent 16-B loads per node visit (result folded into a never-true test)
                for (int r = 0; r < MRT_PROBE_EXTRA_LOADS; r++) {
                    const float4 x = s.wnodes[WNODE_STRIDE * (size_t)((pending * 2654435761u + 977u * (r + 1)) % s.num_wnodes) + (r % 5)];
                    if (x.x == 1.2345e-30f) tri_hits |= 1u;
                }
#endif
#ifdef MRT_PROBE_EXTRA_VALU       // bottleneck probe: extra dependent VALU work per node visit
                { float acc = bx; for (int r = 0; r < MRT_PROBE_EXTRA_VALU; r++) acc = __builtin_fmaf(acc, ax, by); if (acc == 1.2345e-30f) tri_hits |= 1u; }
#endif
                if ((g_mask >> 8) != 0) { wstack_push(stack, sp, lane, g_base, g_mask); sp++; }
                g_base = __float_as_uint(n1.x); g_mask = (node_hits << 8) | imask;
                t_base = __float_as_uint(n1.y); t_mask = tri_hits;
            }
        }
#ifndef MRT_WIDE_TRI_REPEAT
#define MRT_WIDE_TRI_REPEAT 1
#endif
#pragma unroll 1
        for (int rep = 0; rep < MRT_WIDE_TRI_REPEAT; rep++) {
            if (rep > 0 && __ballot(live && t_mask != 0) == 0ull) break;
            if (live && t_mask != 0 && (MRT_WIDE_ORDER == 1 || was_tri)) {
            const bool any = (pixw >> 31) != 0;
                const uint32_t k = (uint32_t)__ffs((int)t_mask) - 1u;
                t_mask &= t_mask - 1u;
                const float4 *__restrict__ pk = s.wpackets + 3 * (size_t)(t_base + k);
                const float4 r0 = pk[0], r1 = pk[1], r2 = pk[2];
                float t, U, V, ad;
                if (tri_test(r0, r1, r2, o, d, 0.0f, h.t, t, U, V, ad)) {
                    const uint32_t gid = __float_as_uint(r0.w);
                    if (any) { h.gid = gid; live = false; unreported = true; }
                    else if (t < h.t || gid < h.gid) { h.t = t; h.gid = gid; best_pk = t_base + k; }
                }
            }
        }
#endif
    }
}

}  // namespace
}  // namespace mrt
