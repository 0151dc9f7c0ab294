// traverse_wide_pool.h — the stream walk of the 8-wide layout (traverse_wide.h) with the TRIANGLE TESTS POOLED ACROSS THE LANES of a wave.
// Third form of Apple's opaque `intersector.intersect` for bounce rays (Raytracing.metal:244, closest) and shadow rays (:367, any); flattened scenes.
//
// Why: in traverse_wide_stream a lane tests its own triangles, one per iteration.  A node visit leaves 0 ... 32 of them (1.9 pending per live lane on DragonScene), so a lane
// spends iterations on triangles only while the node half of the iteration — the expensive one, ~250 VALU instructions paid by the whole wave — runs without it, and the triangle
// half runs for the ~51 % of the lanes that have one: 47 % of the lanes active per VALU instruction (profiles/r04_summary.json, profiles/r05_pooling_probe.txt).  Here a node
// visit POSTS its triangles — {owner lane, packet} words — to a ring in the wave's LDS, and the triangle half of an iteration takes the 64 oldest entries whoever posted them:
// lane i tests entry i with the owner's ray (read from LDS) and folds a hit into the owner's 64-bit key {t bits, triangle id} with one LDS atomic min — minimum t, ties to the
// lowest id: the order-free rule of the other walks, so the image is theirs bit for bit.  The triangle half then runs on full waves, and only when 48 entries have gathered (or
// nothing else is left to do); a lane with nodes left visits one in EVERY iteration, with the limit its key holds by then.
//
// Per wave in LDS (POOL_WORDS 32-bit words in front of the stack): key[64] (u64: t bits << 32 | id; id = ~0: nothing yet), U / V / |det| [3][64] of the hit that holds the
// key, the rays o.xyz d.xyz [6][64], the ring [POOL_Q].  The ring is first in, first out and its head and tail are wave-uniform counters: a lane's ray is
// finished when it has no node left, nothing left to put into the ring and the head has passed its last entry; its slot is then reported and refilled as in the stream walk.
#pragma once
#include "traverse_wide.h"

namespace mrt {
namespace {

#ifndef MRT_POOL_Q
#define MRT_POOL_Q 256            // ring entries (a power of two): four full triangle halves
#endif
#ifndef MRT_POOL_AT
#define MRT_POOL_AT 32            // the triangle half runs when this many entries wait (or when no lane has a node to visit)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MRT_LDS __attribute__((address_space(3)))
#else
#define MRT_LDS
#endif
#define POOL_SYNC() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
#ifndef MRT_POOL_REFILL_AT
#define MRT_POOL_REFILL_AT MRT_WIDE_REFILL_AT      // idle lanes at which the wave reports and refills
#endif
constexpr uint32_t POOL_Q = MRT_POOL_Q;
constexpr uint32_t POOL_WORDS = 128 + 192 + 384 + POOL_Q;          // key, U V |det|, rays, ring
static_assert((POOL_Q & (POOL_Q - 1)) == 0 && POOL_Q >= 128, "the ring's size must be a power of two and hold two triangle halves");

template <class Chunks, class RayFetch, class Emit>
MRT_DEV void traverse_wide_pool(const SceneView &s, Chunks next_chunk, uint32_t *lds /* POOL_WORDS + depth x WIDE_STACK_LEVEL_BYTES / 4 words of this wave */, RayFetch fetch, Emit emit, StreamStats *ss = nullptr) {
    const uint32_t lane = threadIdx.x & 63;
    // LDS pointers BY TYPE (ds_read / ds_write / ds_min_u64): through generic pointers these accesses become flat loads and stores.  What one lane writes another reads in a later
    // phase of the same iteration: the phases are separated by POOL_SYNC (the wave's LDS operations complete in order; the clobber keeps the compiler from caching or moving them).
    typedef MRT_LDS unsigned long long lds_u64; typedef MRT_LDS uint32_t lds_u32; typedef MRT_LDS float lds_f32;
    lds_u64 *const key = (lds_u64 *)lds;
    lds_u32 *const key32 = (lds_u32 *)lds;            // [2 * lane] = id, [2 * lane + 1] = t bits
    lds_f32 *const huv = (lds_f32 *)lds + 128;
    lds_f32 *const rays = (lds_f32 *)lds + 320;          // [6][64]: o.x o.y o.z d.x d.y d.z by lane — six ds_read_b32 per tester, free of bank conflicts whatever the owners (two float4 per lane read with
                                                          // ds_read_b128 put owners 8 apart on the same banks: SQ_LDS_BANK_CONFLICT was a third of the kernel's LDS cycles, profiles/r05_pool_ab.txt)
    lds_u32 *const ring = (lds_u32 *)lds + 704;
    uint32_t *const stack = lds + POOL_WORDS;
    uint32_t q_head = 0, q_tail = 0;                  // wave-uniform, only ever grow; entry k lives in ring[k & (POOL_Q - 1)]
    const unsigned long long lt = (1ull << lane) - 1ull;
    // prefetched batch (as in traverse_wide_stream): rays batch_base .. + batch_n - 1, one per lane; pB.w = tag | any-hit flag << 31
    float4 pA = make_float4(0, 0, 0, 0), pB = pA;
    uint32_t batch_n = 0, batch_used = 0, cur = 0, end = 0;
    bool more = true;
    // the lane's ray
    bool live = false, unreported = false;
    uint32_t tagw = 0;
    f3 o = mk3(0, 0, 0); float ix = 0, iy = 0, iz = 0; bool nx = false, ny = false, nz = false; uint32_t oct = 0;
    uint32_t g_base = 0, g_mask = 0;                  // imask | permuted hit bits << 8 | stack depth << 16
    uint32_t t_base = 0, t_mask = 0;                  // triangles of the last node not yet posted
    uint32_t last_end = 0;                            // ring position behind the last entry of this ray: all of its triangles are tested once the head is there
    uint32_t nxt_e = 0;                               // ring entry q_head + lane, read at the END of an iteration for the next one (no LDS round trip in front of its loads)
    for (;;) {
        // a ray with no node left, everything posted and everything tested is finished
        if (live && t_mask == 0u && (g_mask & 0xFFFFFF00u) == 0u && (int32_t)(q_head - last_end) >= 0) { live = false; unreported = true; }
        const unsigned long long m_idle = __ballot(!live);
        const uint32_t n_idle = (uint32_t)__popcll(m_idle);
        if (n_idle >= (uint32_t)MRT_POOL_REFILL_AT || m_idle == ~0ull) {
            if (unreported) {
                const uint32_t id = key32[2 * lane], tb = key32[2 * lane + 1];
                TravHit h; h.t = __uint_as_float(tb); h.U = huv[lane]; h.V = huv[64u + lane]; h.ad = huv[128u + lane]; h.gid = id;
                emit(tagw & 0x7FFFFFFFu, (tagw >> 31) != 0, id != 0xFFFFFFFFu, h); unreported = false;
            }
            if (batch_used >= batch_n) {
                if (cur >= end && more) more = next_chunk(cur, end);
                batch_n = cur < end ? min(64u, end - cur) : 0u; batch_used = 0;
                if (lane < batch_n) { uint32_t tag = 0, is_any = 0; fetch(cur + lane, pA, pB, tag, is_any); pB.w = __uint_as_float((tag & 0x7FFFFFFFu) | (is_any << 31)); }
                cur += batch_n;
            }
            const uint32_t avail = batch_n - batch_used;
            if (avail == 0) { if (m_idle == ~0ull) break; }
            else {
                const uint32_t rank = (uint32_t)__popcll(m_idle & lt);
                const bool take = !live && rank < avail;
                const int sl = (int)(take ? batch_used + rank : lane);
                const float ax_ = __shfl(pA.x, sl), ay_ = __shfl(pA.y, sl), az_ = __shfl(pA.z, sl), aw_ = __shfl(pA.w, sl);
                const float bx_ = __shfl(pB.x, sl), by_ = __shfl(pB.y, sl), bz_ = __shfl(pB.z, sl), bw_ = __shfl(pB.w, sl);
                if (take) {
                    o = mk3(ax_, ay_, az_); ix = box_inv(bx_); iy = box_inv(by_); iz = box_inv(bz_);
                    nx = bx_ < 0.0f; ny = by_ < 0.0f; nz = bz_ < 0.0f; oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                    tagw = __float_as_uint(bw_);
                    g_base = 0u; g_mask = s.num_wnodes != 0 ? 0x100u : 0u; t_base = 0; t_mask = 0; last_end = q_head;
                    rays[lane] = ax_; rays[64u + lane] = ay_; rays[128u + lane] = az_; rays[192u + lane] = bx_; rays[256u + lane] = by_; rays[320u + lane] = bz_;
                    key32[2 * lane] = 0xFFFFFFFFu; key32[2 * lane + 1] = __float_as_uint(aw_);          // nothing yet; the ray's own limit (+inf for a bounce ray)
                    live = true;
                }
                batch_used += min(avail, n_idle);
                if (ss) { ss->refills++; ss->refill_lanes += min(avail, n_idle); }
                POOL_SYNC();
                continue;
            }
        }
        // ---- this iteration's node per lane: the nearest remaining hit child, or the top of the stack (a lane that still has triangles to post waits for room in the ring)
        bool want_node = live && t_mask == 0u;
        uint32_t pending = 0;
        if (want_node) {
            if ((g_mask & 0xFF00u) == 0u) {
                const uint32_t sp = g_mask >> 16;
                if (sp == 0u) want_node = false;
                else { wstack_pop(stack, sp - 1u, lane, g_base, g_mask); g_mask |= (sp - 1u) << 16; }
            }
            if (want_node) {
                const uint32_t hits = (g_mask >> 8) & 0xFFu;
                const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;
                g_mask &= ~(0x100u << b);
                const uint32_t slot = b ^ oct;
                pending = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));
            }
        }
        // ---- this iteration's triangle per lane: entry q_head + lane of the ring, when enough have gathered or nothing else is left to do
        const uint32_t waiting = q_tail - q_head;
        const bool tri_half = waiting >= (uint32_t)MRT_POOL_AT || (waiting != 0u && __ballot(want_node) == 0ull);          // (wave-uniform)
        const uint32_t n_tri = tri_half ? min(waiting, 64u) : 0u;
        const bool tester = lane < n_tri;
        uint32_t owner = 0, tri_pk = 0;
        if (tester) { owner = nxt_e & 63u; tri_pk = nxt_e >> 6; }
        q_head += n_tri;
        if (ss) { ss->iters++; ss->live_sum += (uint32_t)__popcll(__ballot(live)); ss->tri_sum += n_tri; ss->node_sum += (uint32_t)__popcll(__ballot(want_node)); }
        // ---- one memory round trip: the node (80 B) and the packet (48 B)
        float4 r0, r1, r2, n0, n1, n2, n3, n4;
        asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r2.x), "=v"(r2.y), "=v"(r2.z));
        asm volatile("" : "=v"(n0.x), "=v"(n0.y), "=v"(n0.z), "=v"(n0.w), "=v"(n1.x), "=v"(n1.y), "=v"(n1.z), "=v"(n1.w), "=v"(n2.x), "=v"(n2.y), "=v"(n2.z), "=v"(n2.w));
        asm volatile("" : "=v"(n3.x), "=v"(n3.y), "=v"(n3.z), "=v"(n3.w), "=v"(n4.x), "=v"(n4.y), "=v"(n4.z), "=v"(n4.w));
        r1.w = 0.0f; r2.w = 0.0f;
        if (tester) {
            MRT_BOUND(tri_pk, s.num_wpackets, 2);
            const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)tri_pk;
            r0 = pk[0]; r1 = pk[1]; r2 = pk[2];
        }
        if (want_node) {
            MRT_BOUND(pending, s.num_wnodes, 1);
            const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending;
            n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[WNODE_N4];
        }
        // ---- the triangle half: entry against its owner's ray; a hit is folded into the owner's key (minimum t, ties to the lowest id)
        if (tester) {
            const f3 oo = mk3(rays[owner], rays[64u + owner], rays[128u + owner]), dd = mk3(rays[192u + owner], rays[256u + owner], rays[320u + owner]);
            const float lim = __uint_as_float(key32[2 * owner + 1]);
            float t, U, V, ad;
            if (tri_test(r0, r1, r2, oo, dd, 0.0f, lim, t, U, V, ad)) {
                const unsigned long long mine = ((unsigned long long)__float_as_uint(t + 0.0f) << 32) | (unsigned long long)__float_as_uint(r0.w);          // (+ 0.0f: a distance of -0 must order as 0)
                (void)__hip_atomic_fetch_min(&key[owner], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                asm volatile("" ::: "memory");
                if (key[owner] == mine) { huv[owner] = U; huv[64u + owner] = V; huv[128u + owner] = ad; }          // the key is this hit's (now): its barycentrics go with it.  Another reference of the same triangle writes the same values.
            }
        }
        POOL_SYNC();
        // ---- the node half: eight child boxes against the ray, with the limit the key holds now
        if (want_node) {
            const uint32_t id = key32[2 * lane];
            const float best_t = __uint_as_float(key32[2 * lane + 1]);
            if ((tagw >> 31) != 0 && id != 0xFFFFFFFFu) { g_mask = 0u; t_mask = 0u; }          // a shadow ray that has its occluder: nothing more to visit (what it has posted is still counted off)
            else {
                uint32_t node_hits, tri_hits;
                wide_node_test<MRT_WIDE_SCALED != 0>(n0, n1, n2, n3, n4, o, ix, iy, iz, nx, ny, nz, oct, 0.0f, best_t, node_hits, tri_hits);
                uint32_t sp = g_mask >> 16;
                if ((g_mask & 0xFF00u) != 0u) { wstack_push(stack, sp, lane, g_base, g_mask & 0xFFFFu); sp++; }
                g_base = __float_as_uint(n1.x) & WNODE_BASE_MASK; g_mask = (sp << 16) | (node_hits << 8) | (__float_as_uint(n0.w) >> 24);
                t_base = __float_as_uint(n1.y); t_mask = tri_hits;
            }
        }
        // ---- post the new triangles (and what did not fit before): up to two per lane and trip — the lanes' first ones, then their second ones, each block compacted by ballot — while the ring has room
        for (;;) {
            const unsigned long long m1 = __ballot(t_mask != 0u);
            if (m1 == 0ull) break;
            const uint32_t room = POOL_Q - (q_tail - q_head);
            if (room == 0u) break;
            const uint32_t rest = t_mask & (t_mask - 1u);
            const unsigned long long m2 = __ballot(rest != 0u);
            const uint32_t c1 = (uint32_t)__popcll(m1), r1 = (uint32_t)__popcll(m1 & lt), r2 = c1 + (uint32_t)__popcll(m2 & lt);
            if (t_mask != 0u && r1 < room) {
                ring[(q_tail + r1) & (POOL_Q - 1u)] = lane | ((t_base + (uint32_t)__ffs((int)t_mask) - 1u) << 6);
                t_mask = rest; last_end = q_tail + r1 + 1u;
                if (rest != 0u && r2 < room) { ring[(q_tail + r2) & (POOL_Q - 1u)] = lane | ((t_base + (uint32_t)__ffs((int)rest) - 1u) << 6); t_mask = rest & (rest - 1u); last_end = q_tail + r2 + 1u; }
            }
            q_tail += min(c1 + (uint32_t)__popcll(m2), room);
        }
        POOL_SYNC();
        // for the next iteration, already now: the ring entry this lane would test
        if (lane < q_tail - q_head) nxt_e = ring[(q_head + lane) & (POOL_Q - 1u)];
    }
}

}  // namespace
}  // namespace mrt
