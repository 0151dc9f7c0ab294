// two_level_passes.h — the binned walk of two-level scenes: TLAS pass + BLAS pass over (ray, instance) pairs (renderer option tl_pairs; docs/HISTORY.md §6.72).
// Replaces `intersector<triangle_data, instancing>::intersect` (Raytracing.metal:244, :367) for scenes committed with instancing = 1 (Renderer.swift:193-213).
#pragma once
#include "traverse_wide.h"
#include "traverse_instanced.h"

namespace mrt {
namespace {

// In the one-loop walk of a two-level scene (k_trace_mixed_wide_persist<true>) the lanes of a wave are out of step — node / triangle / level change — and dragon x 4 costs
// 17.1 wave-iterations per 64 bounce rays where the flattened scene costs 13.5 (tools/archive/two_level_binning_probe.py).  Here the bounce / shadow rays of a shade pass take two launches:
//   k_tl_top    the same loop, but an instance of more than eight triangles is not entered: {ray, instance} goes to a queue (ballot-compacted; a wave reserves 256 slots per atomic: PairQueue).
//               What a ray finds at the TLAS level (walls, floor: tested in place) becomes its result so far: a 64-bit key (t bits << 32 | global triangle id), ~0 = nothing;
//               a shadow ray not occluded so far sets its pixel's byte.
//   k_tl_blas   every pair walked in its instance's object space by the FLATTENED loop (per-ray root; the ray transformed once, at fetch), starting with the distance its ray
//               has so far; a hit is folded into the ray's key with atomicMin — minimum t, ties to the lowest global id: the one-loop walk's result — or clears the pixel's byte.
// k_shade<.., PAIRS> of the next bounce turns a key back into a hit record (the winning triangle re-tested in object space: the traversal's own arithmetic).
__global__ void __launch_bounds__(64, MRT_TWO_LEVEL_WAVES) k_tl_top(SceneView s, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, unsigned long long *__restrict__ keys,
                                                                 const float4 *__restrict__ srayA, const float4 *__restrict__ srayB, const unsigned long long *__restrict__ counts,
                                                                 uint32_t *__restrict__ work, uint32_t chunk, uint8_t *__restrict__ lit, uint4 *__restrict__ pairs, uint32_t *__restrict__ pair_count, uint32_t pair_cap, uint32_t stack_words) {
    extern __shared__ uint32_t stk_dyn[];
    const unsigned long long c = *counts;
    const uint32_t n_next = (uint32_t)c, n_shadow = (uint32_t)(c >> 32), n = n_next + n_shadow;
    if (blockIdx.x * chunk >= n) return;
    uint32_t *const cursor = stk_dyn + stack_words;          // two words behind the wave's stack: its block of the pair queue
    cursor[0] = 0; cursor[1] = 0;
    const PairQueue pq{pairs, pair_count, pair_cap, cursor};
    traverse_wide_stream<true, false, false, PairQueue>(s, SharedCounter{work, n, chunk}, stk_dyn,
        [&](uint32_t i, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) {
            const bool sh = i >= n_next; tag = sh ? i - n_next : i; is_any = sh ? 1u : 0u;
            A = qload(sh ? &srayA[tag] : &rayA[tag]); B = qload(sh ? &srayB[tag] : &rayB[tag]);
            if (!sh) A.w = __builtin_inff();          // a bounce ray's tmax word carries the throughput chain
            else tag = __float_as_uint(B.w);            // a shadow ray reports to its pixel's byte
        },
        [&](uint32_t j, bool is_any, bool hit, const TravHit &h) {
            if (is_any) { if (!hit) lit[4 * (size_t)j] = 1; }          // not occluded at the TLAS level: lit unless one of its pairs finds an occluder
            else {
                __builtin_nontemporal_store(hit ? ((unsigned long long)__float_as_uint(h.t + 0.0f) << 32) | h.gid : ~0ull, &keys[j]);      // (+ 0.0f: a distance of -0 must order as 0)
            }
        }, nullptr, pq);
    pq.close();
}
// The TLAS pass of a scene with FEW instances (at most TL_FLAT_MAX_INSTANCES; DragonScene x 4 has ten), without a tree: one ray per lane, every lane visits every instance in the same
// order — the rows and boxes are wave-uniform (scalar loads), nothing diverges, nothing is gathered, no stack, no refill.  First the instances of at most eight triangles, tested in
// place in object space (they give the ray its bound); then the large ones: the ray against the instance's BLAS box in object space, and a pair for the BLAS pass where it enters
// before that bound.  A refused pair (queue full) is walked here, one ray per lane (traverse_wide from the BLAS root).  Pairs leave instance-major, so the BLAS pass's waves see one
// instance at a time.  (Compacting the (ray, instance) candidates across the wave — world boxes first, then a list in LDS taken 64 entries at a time — measured -12 % in round 6:
// per-lane gathers of rows and packets cost more than the idle lanes they save; profiles/r06_two_level_ab.txt D.)  The stream walk of the 8-wide TLAS (k_tl_top) spent 1.16 ms per launch on the 12 M rays of an 8-frame pass of dragon x 4, refilling lanes every other iteration; this takes them in 0.53 ms (profiles/r04_two_level_binned_ab.txt).
constexpr uint32_t TL_FLAT_MAX_INSTANCES = 64;
// ONE ray of that pass (every lane of the wave calls it; `done` = the lane has no ray, or its shadow ray is already occluded).  push(id): the ray enters large instance id —
// true = a pair for the BLAS pass is (or will be) queued; false = walk(I, oo, dd, bound, h) walks the instance in place.  Leaves the closest hit among the small instances (and refused pairs) in
// best_t / best_gid, or done = true for an occluded shadow ray.
template <class Push, class WalkInPlace>
MRT_DEV void tl_flat_ray(const SceneView &s, const f3 o, const f3 d, const bool sh, float &best_t, uint32_t &best_gid, bool &done, Push push, WalkInPlace walk) {
    for (uint32_t id = 0; id < s.num_inst; id++) {                  // the small instances, in place
        const InstanceDev &I = s.inst[id];
        if (I.ntri == 0u || I.ntri > 8u || s.inst_box[4 * id].x > s.inst_box[4 * id + 1].x) continue;          // (wave-uniform)
        const f3 oo = to_object_point(I, o), dd = to_object_dir(I, d);
        for (uint32_t k = 0; k < I.ntri; k++) {
            const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)(I.packet_base + k);
            const float4 q0 = pk[0], q1 = pk[1], q2 = pk[2];
            float t, U, V, ad;
            if (!done && tri_test(q0, q1, q2, oo, dd, 0.0f, best_t, t, U, V, ad)) {
                const uint32_t gid = I.gid_base + __float_as_uint(q0.w);
                if (sh) done = true;                                // occluded: lit stays 0
                else if (t < best_t || gid < best_gid) { best_t = t; best_gid = gid; }      // (t <= best_t here: ties go to the lowest global id)
            }
        }
    }
    for (uint32_t id = 0; id < s.num_inst; id++) {                  // the large instances: a pair where the ray enters the BLAS's box before its bound
        const InstanceDev &I = s.inst[id];
        const float4 blo = s.inst_box[4 * id], bhi = s.inst_box[4 * id + 1];
        if (I.ntri <= 8u || blo.x > bhi.x) continue;
        const f3 oo = to_object_point(I, o), dd = to_object_dir(I, d);
        const float ix = box_inv(dd.x), iy = box_inv(dd.y), iz = box_inv(dd.z);
        const bool enters = !done && rope_box_hit(blo, bhi, ix, iy, iz, -(oo.x * ix), -(oo.y * iy), -(oo.z * iz), 0.0f, best_t);
        if (enters) {
            if (!push(id)) {                    // the queue is full: this instance is walked here
                TravHit h;
                if (walk(I, oo, dd, best_t, h)) {
                    if (sh) done = true;
                    else { const uint32_t gid = I.gid_base + h.gid; if (h.t < best_t || gid < best_gid) { best_t = h.t; best_gid = gid; } }
                }
            }
        }
    }
}
__global__ void __launch_bounds__(64) k_tl_top_flat(SceneView s, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, unsigned long long *__restrict__ keys,
                                                    const float4 *__restrict__ srayA, const float4 *__restrict__ srayB, const unsigned long long *__restrict__ counts,
                                                    uint8_t *__restrict__ lit, uint4 *__restrict__ pairs, uint32_t *__restrict__ pair_count, uint32_t pair_cap, uint32_t stack_words) {
    extern __shared__ uint32_t stk_dyn[];
    const unsigned long long c = *counts;
    const uint32_t n_next = (uint32_t)c, n = n_next + (uint32_t)(c >> 32);
    if (blockIdx.x * 64u >= n) return;
    uint32_t *const cursor = stk_dyn + stack_words;
    cursor[0] = 0; cursor[1] = 0;
    const PairQueue pq{pairs, pair_count, pair_cap, cursor};
    for (uint32_t base = blockIdx.x * 64u; base < n; base += gridDim.x * 64u) {          // (wave-uniform) the launch has as many waves as the chip has slots for them
        const uint32_t i = base + threadIdx.x;
        const bool active = i < n, sh = i >= n_next;
        const uint32_t j = sh ? i - n_next : i;
        float4 A = make_float4(0.0f, 0.0f, 0.0f, -1.0f), B = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (active) { A = qload(sh ? &srayA[j] : &rayA[j]); B = qload(sh ? &srayB[j] : &rayB[j]); if (!sh) A.w = __builtin_inff(); }
        const uint32_t tagw = sh ? (__float_as_uint(B.w) | 0x80000000u) : j;      // what the BLAS pass reports to: the pixel's byte (shadow) / the ray's key
        float best_t = A.w; uint32_t best_gid = 0xFFFFFFFFu; bool done = !active;
        tl_flat_ray(s, mk3(A), mk3(B), sh, best_t, best_gid, done, [&](uint32_t id) { return pq.push(i, id, best_t, tagw); },
                    [&](const InstanceDev &I, const f3 oo, const f3 dd, float bound, TravHit &h) { return traverse_wide<false, false, true>(s, oo, dd, 0.0f, bound, h, stk_dyn, nullptr, sh, I.wroot); });
        if (active) {
            if (sh) { if (!done) lit[4 * (size_t)(tagw & 0x7FFFFFFFu)] = 1; }          // not occluded so far: lit unless one of its pairs finds an occluder
            else __builtin_nontemporal_store(best_gid != 0xFFFFFFFFu ? ((unsigned long long)__float_as_uint(best_t + 0.0f) << 32) | best_gid : ~0ull, &keys[j]);
        }
    }
    pq.close();
}
// The BLAS pass: the pairs name their rays by index in the combined queue [bounce rays | shadow rays].
__global__ void __launch_bounds__(64, MRT_WIDE_STREAM_WAVES) k_tl_blas(SceneView s, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, unsigned long long *__restrict__ keys,
                                                                   const float4 *__restrict__ srayA, const float4 *__restrict__ srayB, const unsigned long long *__restrict__ counts,
                                                                   uint32_t *__restrict__ work, uint32_t chunk, uint8_t *__restrict__ lit, const uint4 *__restrict__ pairs, const uint32_t *__restrict__ pair_count, uint32_t pair_cap) {
    extern __shared__ uint32_t stk_dyn[];
    // whole blocks only (blocks beyond the capacity were refused — their rays walked in place — but counted)
    const uint32_t n_next = (uint32_t)*counts, np = min(*pair_count, pair_cap / PairQueue::BLOCK * PairQueue::BLOCK);
    if (blockIdx.x * chunk >= np) return;
    traverse_wide_stream<false, false, true>(s, SharedCounter{work, np, chunk}, stk_dyn,
        [&](uint32_t k, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any, uint32_t &root) {
            const float4 Pf = qload(reinterpret_cast<const float4 *>(&pairs[k]));
            const uint32_t i = __float_as_uint(Pf.x), id = __float_as_uint(Pf.y);
            if (i == 0xFFFFFFFFu) {          // the unused rest of a wave's block: a ray that cannot hit anything
                A = make_float4(0.0f, 0.0f, 0.0f, -1.0f); B = make_float4(0.0f, 0.0f, 1.0f, 0.0f); tag = k; is_any = 1u; root = 0u;
                return;
            }
            const bool sh = i >= n_next; const uint32_t j = sh ? i - n_next : i;
            const float4 Aw = qload(sh ? &srayA[j] : &rayA[j]), Bw = qload(sh ? &srayB[j] : &rayB[j]);
            const InstanceDev &I = s.inst[id];
            const f3 o = to_object_point(I, mk3(Aw)), d = to_object_dir(I, mk3(Bw));        // direction not renormalised: t stays the world distance
            float tmax = Aw.w;
            if (!sh) { const unsigned long long key = keys[j]; tmax = key == ~0ull ? __builtin_inff() : __uint_as_float((uint32_t)(key >> 32)); }      // what the ray has so far (TLAS-level hits; other pairs of the same ray may shorten it further while this one walks)
            A = make_float4(o.x, o.y, o.z, tmax); B = make_float4(d.x, d.y, d.z, 0.0f);
            tag = k; is_any = sh ? 1u : 0u; root = I.wroot;
        },
        [&](uint32_t k, bool is_any, bool hit, const TravHit &h) {
            if (!hit) return;
            const uint4 P = pairs[k];
            if (is_any) lit[4 * (size_t)(P.w & 0x7FFFFFFFu)] = 0;          // occluded inside this instance
            else {
                const uint32_t gid = s.inst[P.y].gid_base + h.gid;      // (h.gid: the triangle's id inside its BLAS)
                atomicMin(&keys[P.x], ((unsigned long long)__float_as_uint(h.t + 0.0f) << 32) | (unsigned long long)gid);          // (a bounce ray's P.x is its index in its own queue either way)
            }
        });
}

}  // namespace
}  // namespace mrt
