// renderer.hip — the per-frame hot path as CDNA4 kernels + the host frame driver.
//
// Replaces `raytracingKernel` (Raytracing.metal:156-405) and the frame driver around it
// (Renderer.draw / update / updateUniforms / createTextures, Renderer.swift:216-357).
// The reference is one megakernel over Apple's opaque intersector; here the frame is a wavefront
// pipeline over ray / hit records in HBM:
//
//   primary rays  Halton-jittered, traced to their closest hit                                (Raytracing.metal:171-247)
//   k_shade       normal interpolation, light pick + evaluation, throughput, NEE shadow-ray emit,
//                 cosine-hemisphere bounce; wave-ballot compaction of both output queues      (:249-391)
//   traversal     bounce rays (closest hit) and shadow rays (any hit) of that shade, ONE launch (:244, :367)
//   accumulate    running average with the previous target                                    (:394-403)
//
// The DEFAULT pass carries up to eight frames and runs
//   k_shade<.., TRACE0>          primary rays generated, traced (last frame's hit as a hint) and shaded in one launch
//   k_trace_mixed_wide_persist   per bounce: its bounce rays (closest hit) and shadow rays (any hit) in ONE launch of persistent waves on the
//                                8-wide layout (traverse_wide.h); a shadow ray that gets through sets one byte
//   k_shade                      bounces 1, 2: the light's contribution goes to a per-bounce plane, the throughput is rebuilt from resource slots
//   k_accumulate_planes          contributions whose byte is set, summed in bounce order; running average (the last passes of a draw: one launch)
// on up to six streams; instanced scenes walk both levels with the same kernels (<TWO_LEVEL>).  Scenes without the 8-wide layout (scene option wide = 0, a tree
// deeper than WIDE_STACK_MAX) and the materials extension / more than three bounces take the general form of the same pipeline (k_trace_primary, k_trace_mixed, k_accumulate).
//
// Record layout (all 16-byte lanes, one dwordx4 per lane per access, fully coalesced):
//   rayA = {origin.xyz, tmax}   rayB = {direction.xyz, pixel}   thr = {throughput.rgb, -}   (path rays)
//   hit  = {t, U/|det|, V/|det|, gid}                                                        (16 B)
//   shadow rays reuse rayA/rayB with tmax = lightDistance - 1e-3; con = {Lc*throughput, -}
#include <hip/hip_ext.h>
#include "renderer.h"
#include "device_math.h"
#include "traverse.h"
#include "traverse_wide.h"
#include "traverse_instanced.h"
#include "shade.h"
#include "two_level_passes.h"
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace mrt {
namespace {

__global__ void k_halton_table(float *__restrict__ tab, uint32_t w0, uint32_t n) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) tab[MRT_HALTON_INTERLEAVED ? (size_t)j * HALTON_TAB_DIMS + blockIdx.y : (size_t)blockIdx.y * n + j] = halton_dev((int)(w0 + j), (int)blockIdx.y + 1);
}

// seeds[s * n + i] = hash(seed, i) + s: sub-frame s of a batch reads its Halton offset with the sub-frame index already added
__global__ void k_seed(uint32_t *__restrict__ seeds, uint32_t n, uint32_t seed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) seeds[(size_t)blockIdx.y * n + i] = seed_hash_dev(seed, i) + blockIdx.y;
}

// the renderer's seed table: by sample index (sub-frame * capacity + slot of this shard), seeds[...] = hash(seed, pixel of the slot) + sub-frame
__global__ void k_seed_slots(uint32_t *__restrict__ seeds, FrameParams fp, uint32_t seed) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= fp.capacity) return;
    int x, y;
    seeds[(size_t)blockIdx.y * fp.capacity + slot] = slot_to_pixel(fp, slot, x, y) ? seed_hash_dev(seed, (uint32_t)y * (uint32_t)fp.width + (uint32_t)x) + blockIdx.y : 0u;
}

// ------------------------------------------------------------------ traversal launches of the pipeline
// Fewer, fatter traversal launches: every launch ends in a latency-bound tail (a handful of waves walking
// the longest rays), so the frame runs  primary -> shade -> [shadow(b) + extend(b+1)] -> shade -> ... .
//   k_trace_primary : primary-ray generation (Raytracing.metal:171-221) fused with the first closest-hit query
//   k_trace_mixed   : one launch over two queues — the next-bounce rays (closest hit -> hit records) and the
//                     shadow rays of the same shade pass (any hit -> sample accumulation).  The two kinds share
//                     the loop; a shadow lane simply stops at its first hit.
// WIDE (flattened scenes with the 8-wide layout): one ray per lane on that layout (traverse_wide_lane), the wave's stack in dynamic LDS; the hint then names a packet of wpackets
template <bool TWO_LEVEL, bool WIDE = false>
__global__ void __launch_bounds__(64) k_trace_primary(SceneView s, FrameParams fp, const uint32_t *__restrict__ seeds, float4 *__restrict__ hits, float4 *__restrict__ dirs, uint32_t *__restrict__ hint) {
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x, sub = blockIdx.y;      // grid = (local tiles, sub-frames of the batch)
    float4 *__restrict__ hits_s = hits + (size_t)sub * fp.capacity;
    int x, y;
    if (!slot_to_pixel(fp, slot, x, y)) {
        if ((int)(slot >> 6) < fp.tiles_local) hits_s[slot] = make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
        return;
    }
    f3 org, dir;
    primary_ray(fp, seeds, sub * fp.capacity + slot, x, y, org, dir);
    // the direction goes to HBM (16 B per pixel: the memory system has headroom, the vector ALUs do not — k_shade reads it back
    // instead of repeating two Halton values, two divisions and a normalisation per pixel)
    qstore(&dirs[(size_t)sub * fp.capacity + slot], make_float4(dir.x, dir.y, dir.z, __uint_as_float(sub * fp.capacity + slot)));
    TravHit h;
    bool hit;
    if (WIDE && !TWO_LEVEL) {
        extern __shared__ uint32_t stk_dyn[];
        if (hint != nullptr) {
            const uint32_t pixel = (uint32_t)y * (uint32_t)fp.width + (uint32_t)x;
            const uint32_t guess = hint[pixel];
            float t0 = __builtin_inff(); uint32_t seed = 0xFFFFFFFFu;
            if (guess < s.num_wpackets) {
                const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)guess;
                float t, U, V, ad;
                if (tri_test(pk[0], pk[1], pk[2], org, dir, 0.0f, __builtin_inff(), t, U, V, ad)) { t0 = t; seed = guess; }
            }
            hit = traverse_wide_lane<true>(s, org, dir, t0, seed, h, stk_dyn);
            if (h.pk != guess) hint[pixel] = h.pk;
        }
        else hit = traverse_wide_lane<false>(s, org, dir, __builtin_inff(), 0xFFFFFFFFu, h, stk_dyn);
    }
    else if (!TWO_LEVEL && hint != nullptr) {
        // the triangle this pixel hit in an earlier frame is tested first: the jittered ray most often hits it again, and the walk then starts with
        // the right distance bound instead of discovering it.  Any packet is a legal guess (a wrong one is one wasted test); the result is unchanged.
        const uint32_t pixel = (uint32_t)y * (uint32_t)fp.width + (uint32_t)x;
        const uint32_t guess = hint[pixel];
        h.t = __builtin_inff(); h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu; h.pk = 0xFFFFFFFFu;
        if (guess < s.num_tris) {
            const float4 *__restrict__ pk = s.packets + 3 * (size_t)guess;
            const float4 q0 = pk[0];
            float t, U, V, ad;
            if (tri_test(q0, pk[1], pk[2], org, dir, 0.0f, __builtin_inff(), t, U, V, ad)) { h.t = t; h.U = U; h.V = V; h.ad = ad; h.gid = __float_as_uint(q0.w); h.pk = guess; }
        }
        hit = traverse<false, false, false, true>(s, org, dir, 0.0f, h.t, h);
        if (h.pk != guess) hint[pixel] = h.pk;
    }
    else hit = TWO_LEVEL ? traverse_instanced<false>(s, org, dir, 0.0f, __builtin_inff(), h) : traverse<false>(s, org, dir, 0.0f, __builtin_inff(), h);
    qstore(&hits_s[slot], hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)));
}

template <bool TWO_LEVEL>
__global__ void __launch_bounds__(64) k_trace_mixed(SceneView s, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, float4 *__restrict__ hits,
                                                    const float4 *__restrict__ srayA, const float4 *__restrict__ srayB, const float4 *__restrict__ scon,
                                                    const unsigned long long *__restrict__ counts, float4 *__restrict__ sample) {
    const unsigned long long c = *counts;
    const uint32_t n_next = (uint32_t)c, n_shadow = (uint32_t)(c >> 32);
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_next + n_shadow) return;
    const bool shadow = i >= n_next;
    const uint32_t j = shadow ? i - n_next : i;
    float4 A = shadow ? srayA[j] : rayA[j]; const float4 B = shadow ? srayB[j] : rayB[j];
    if (!shadow) A.w = __builtin_inff();          // a bounce ray's tmax word may carry the throughput chain
    TravHit h;
    bool hit = TWO_LEVEL ? traverse_instanced<false, true>(s, mk3(A), mk3(B), 0.0f, A.w, h, shadow) : traverse<false, false, true>(s, mk3(A), mk3(B), 0.0f, A.w, h, nullptr, shadow);
    if (shadow) {
        if (!hit) {
            uint32_t pix = __float_as_uint(B.w);
            float4 cc = scon[j], a = sample[pix];
            sample[pix] = make_float4(a.x + cc.x, a.y + cc.y, a.z + cc.z, 0.0f);   // one shadow ray per pixel per bounce: no atomics
        }
    } else {
        hits[j] = hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
    }
}

// ------------------------------------------------------------------ 8-wide layout (LDS stack, traverse_wide.h)
// One wave walks `rays_per_wave` consecutive rays of the combined queue
// [next-bounce rays | shadow rays] with lane refill (traverse_wide_stream).  Longer ranges amortise the drain at the end
// of a wave's range, shorter ones keep the launch's tail short and the grid large; the host picks the range from the
// size of the launch (stream_rays_per_wave): one frame per pass (4 M slots): 384 measured best (256: -2 %, 512: -1 %,
// 1024: -7 %); four frames per pass (17 M slots): 1024 (384: -4.5 %, 768: -1.5 %, 2048: equal on the full frame, -3 % on
// primary + shadow).
#ifdef MRT_WIDE_STREAM_RAYS
constexpr uint32_t WIDE_STREAM_RAYS_FIXED = MRT_WIDE_STREAM_RAYS;     // tuning builds
#else
constexpr uint32_t WIDE_STREAM_RAYS_FIXED = 0;
#endif
static inline uint32_t stream_rays_per_wave(size_t slots) {
    if (WIDE_STREAM_RAYS_FIXED) return WIDE_STREAM_RAYS_FIXED;
    const size_t batches = slots / 64 / 10240;                             // 64-ray batches per wave if the launch had ~10 K waves
    return 64u * (uint32_t)std::min<size_t>(16, std::max<size_t>(6, batches));
}
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(64, TWO_LEVEL ? MRT_TWO_LEVEL_WAVES : MRT_WIDE_STREAM_WAVES) k_trace_mixed_wide_stream(SceneView s, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, float4 *__restrict__ hits,
                                                                const float4 *__restrict__ srayA, const float4 *__restrict__ srayB, const float4 *__restrict__ scon,
                                                                const unsigned long long *__restrict__ counts, float4 *__restrict__ sample, uint32_t rays_per_wave, uint8_t *__restrict__ lit, uint32_t even_waves) {
    extern __shared__ uint32_t stk_dyn[];
    const unsigned long long c = *counts;
    const uint32_t n_next = (uint32_t)c, n_shadow = (uint32_t)(c >> 32), n = n_next + n_shadow;
    // even_waves (renderer option stream_even): the launch has that many waves and the rays the queue really holds are split evenly among them (whole 64-ray
    // batches) instead of rays_per_wave each to the first n / rays_per_wave waves of a grid sized for the queue's capacity
    if (even_waves) rays_per_wave = max(64u, ((n + even_waves - 1u) / even_waves + 63u) & ~63u);
    const uint32_t begin = blockIdx.x * rays_per_wave;
    if (begin >= n) return;
    traverse_wide_stream<TWO_LEVEL>(s, OneRange{begin, min(n, begin + rays_per_wave)}, stk_dyn,
        [&](uint32_t i, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) {      // tag = index in the ray's own queue
            const bool sh = i >= n_next; tag = sh ? i - n_next : i; is_any = sh ? 1u : 0u;
            A = qload(sh ? &srayA[tag] : &rayA[tag]); B = qload(sh ? &srayB[tag] : &rayB[tag]);
            if (!sh) A.w = __builtin_inff();          // a bounce ray's tmax word may carry the throughput chain
            else if (lit) tag = __float_as_uint(B.w);   // shadow planes: the ray reports to its pixel's byte
        },
        [&](uint32_t j, bool is_any, bool hit, const TravHit &h) {
            if (is_any) {
                if (!hit) {
                    if (lit) lit[4 * (size_t)j] = 1;          // (lit points at this bounce's byte of pixel 0: four bytes per pixel and frame)
                    else { const uint32_t pix = __float_as_uint(srayB[j].w); float4 cc = qload(&scon[j]), a = q2load(&sample[pix]); q2store(&sample[pix], make_float4(a.x + cc.x, a.y + cc.y, a.z + cc.z, 0.0f)); }
                }
            } else {
                qstore(&hits[j], hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)));
            }
        });
}

#ifdef MRT_WAVE_TIMES      // diagnostics build (tools/archive/wave_times.py): when does every wave of the first traversal launch of a pass start and end?
__device__ unsigned long long g_wave_times[2 * 8192];
__device__ uint32_t g_wave_iters[4 * 8192];      // per wave of that launch: iterations | drain iterations + longest iteration << 12 | live lanes summed over the drain iterations | drain start tick
__device__ unsigned long long g_drain_probe[2][9];      // StreamStats::dr_*, summed over the waves of that launch: [at most 16 | at most 4 live lanes][6 stack-depth bins, hit children pending, triangles pending, samples]
#endif
// Persistent variant: the grid is the number of wave slots of the chip (or fewer for a small queue) and every wave pulls
// `chunk` consecutive rays of the combined queue at a time from `work` (zeroed by k_accumulate at the end of the previous pass).
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(64, TWO_LEVEL ? MRT_TWO_LEVEL_WAVES : MRT_WIDE_STREAM_WAVES) k_trace_mixed_wide_persist(SceneView s, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, float4 *__restrict__ hits,
                                                                const float4 *__restrict__ srayA, const float4 *__restrict__ srayB, const float4 *__restrict__ scon,
                                                                const unsigned long long *__restrict__ counts, float4 *__restrict__ sample, uint32_t *__restrict__ work, uint32_t chunk, uint8_t *__restrict__ lit, uint32_t subframes /* > 0: XCD regions (XcdRegions), the pass's sub-frames */) {
    extern __shared__ uint32_t stk_dyn[];
    const unsigned long long c = *counts;
    const uint32_t n_next = (uint32_t)c, n_shadow = (uint32_t)(c >> 32), n = n_next + n_shadow;
#ifdef MRT_WAVE_TIMES
    const uint32_t wt_tag = chunk >> 24; chunk &= 0xFFFFFFu;
    const unsigned long long wt0 = wall_clock64();
    StreamStats wst{0, 0, 0, 0, 0, 0};
    struct WT { unsigned long long t0; uint32_t tag; StreamStats &st; __device__ ~WT() { if ((threadIdx.x & 63) == 0 && tag == 0 && blockIdx.x < 8192) {
        g_wave_times[2 * blockIdx.x] = t0; g_wave_times[2 * blockIdx.x + 1] = wall_clock64();
        g_wave_iters[4 * blockIdx.x] = st.iters | (st.drain_le8 << 16); g_wave_iters[4 * blockIdx.x + 1] = st.drain_iters | (st.maxdt << 12); g_wave_iters[4 * blockIdx.x + 2] = st.drain_live; g_wave_iters[4 * blockIdx.x + 3] = (uint32_t)st.drain_t0; } } } wt{wt0, wt_tag, wst};
#endif
    if (blockIdx.x * chunk >= n) return;            // more waves than chunks (small queue): the surplus leaves at once
    auto fetch = [&](uint32_t i, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) {
            const bool sh = i >= n_next; tag = sh ? i - n_next : i; is_any = sh ? 1u : 0u;
            A = qload(sh ? &srayA[tag] : &rayA[tag]); B = qload(sh ? &srayB[tag] : &rayB[tag]);
            if (!sh) A.w = __builtin_inff();          // a bounce ray's tmax word may carry the throughput chain
            else if (lit) tag = __float_as_uint(B.w);   // shadow planes: the ray reports to its pixel's byte
        };
    auto emit = [&](uint32_t j, bool is_any, bool hit, const TravHit &h) {
            if (is_any) {
                if (!hit) {
                    if (lit) lit[4 * (size_t)j] = 1;          // (lit points at this bounce's byte of pixel 0: four bytes per pixel and frame)
                    else { const uint32_t pix = __float_as_uint(srayB[j].w); float4 cc = qload(&scon[j]), a = q2load(&sample[pix]); q2store(&sample[pix], make_float4(a.x + cc.x, a.y + cc.y, a.z + cc.z, 0.0f)); }
                }
            } else {
                qstore(&hits[j], hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)));
            }
        };
#ifdef MRT_WAVE_TIMES
    StreamStats *const wss = wt_tag == 0 ? &wst : nullptr;
#else
    StreamStats *const wss = nullptr;
#endif
    // two-level scenes keep the one counter (the first of the eight): the per-XCD form measured no gain there and costs the in-loop walk five spilled registers
    if constexpr (TWO_LEVEL) traverse_wide_stream<true>(s, SharedCounter{work, n, chunk}, stk_dyn, fetch, emit, wss);
    else traverse_wide_stream<false>(s, XcdRegions{work, n_next, n, chunk, subframes, blockIdx.x & 7u}, stk_dyn, fetch, emit, wss);
}

// The pulling launch of flattened scenes with the hit words in LDS (traverse_wide.h StreamExt; renderer option hit_lds, default): every wave has its own hit
// words (1 KB + the root's hit bits of the prefetched batch) in front of its stack.  Shadow planes only (`lit`): the default path of the pipeline.
__global__ void __launch_bounds__(64, MRT_WIDE_STREAM_WAVES) k_trace_mixed_wide_persist_x(SceneView s, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, float4 *__restrict__ hits,
                                                                const float4 *__restrict__ srayA, const float4 *__restrict__ srayB, const unsigned long long *__restrict__ counts,
                                                                uint32_t *__restrict__ work, uint32_t chunk, uint8_t *__restrict__ lit, uint32_t subframes) {
    extern __shared__ uint32_t lds_dyn[];
    const unsigned long long c = *counts;
    const uint32_t n_next = (uint32_t)c, n_shadow = (uint32_t)(c >> 32), n = n_next + n_shadow;
    if (blockIdx.x * chunk >= n) return;            // more waves than chunks (small queue): the surplus leaves at once
    auto fetch = [&](uint32_t i, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) {
            const bool sh = i >= n_next; tag = sh ? i - n_next : i; is_any = sh ? 1u : 0u;
            A = qload(sh ? &srayA[tag] : &rayA[tag]); B = qload(sh ? &srayB[tag] : &rayB[tag]);
            if (!sh) A.w = __builtin_inff();          // a bounce ray's tmax word may carry the throughput chain
            else tag = __float_as_uint(B.w);          // shadow planes: the ray reports to its pixel's byte
        };
    auto emit = [&](uint32_t j, bool is_any, bool hit, const TravHit &h) {
            if (is_any) { if (!hit) lit[4 * (size_t)j] = 1; }
            else qstore(&hits[j], hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)));
        };
    StreamExt<true> ext{reinterpret_cast<float *>(lds_dyn), lds_dyn + 256u};
    traverse_wide_stream<false, false, false, NoPairs, StreamExt<true>>(s, XcdRegions{work, n_next, n, chunk, subframes, blockIdx.x & 7u}, lds_dyn + HIT_LDS_WORDS, fetch, emit, nullptr, NoPairs{}, ext);
}

// The static split of small launches (k_trace_mixed_wide_stream) with the hit words in LDS (renderer option hit_lds): flattened scenes, shadow planes.
__global__ void __launch_bounds__(64, MRT_WIDE_STREAM_WAVES) k_trace_mixed_wide_stream_x(SceneView s, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, float4 *__restrict__ hits,
                                                                const float4 *__restrict__ srayA, const float4 *__restrict__ srayB, const unsigned long long *__restrict__ counts,
                                                                uint32_t rays_per_wave, uint8_t *__restrict__ lit, uint32_t even_waves) {
    extern __shared__ uint32_t lds_dyn[];
    const unsigned long long c = *counts;
    const uint32_t n_next = (uint32_t)c, n_shadow = (uint32_t)(c >> 32), n = n_next + n_shadow;
#ifdef MRT_WAVE_TIMES
    const uint32_t wt_tag = rays_per_wave >> 24; rays_per_wave &= 0xFFFFFFu;
    StreamStats wst{0, 0, 0, 0, 0, 0};
    struct WT { unsigned long long t0; uint32_t tag; StreamStats &st; __device__ ~WT() { if ((threadIdx.x & 63) == 0 && tag == 0 && blockIdx.x < 8192) {
        g_wave_times[2 * blockIdx.x] = t0; g_wave_times[2 * blockIdx.x + 1] = wall_clock64();
        g_wave_iters[4 * blockIdx.x] = st.iters | (st.drain_le8 << 16); g_wave_iters[4 * blockIdx.x + 1] = st.drain_iters | (st.maxdt << 12); g_wave_iters[4 * blockIdx.x + 2] = st.drain_live; g_wave_iters[4 * blockIdx.x + 3] = (uint32_t)st.drain_t0; }
        if ((threadIdx.x & 63) == 0 && tag == 0) for (int c_ = 0; c_ < 2; c_++) { for (int d_ = 0; d_ < 6; d_++) atomicAdd(&g_drain_probe[c_][d_], (unsigned long long)st.dr_hist[c_][d_]);
            atomicAdd(&g_drain_probe[c_][6], (unsigned long long)st.dr_kids[c_]); atomicAdd(&g_drain_probe[c_][7], (unsigned long long)st.dr_tris[c_]); atomicAdd(&g_drain_probe[c_][8], (unsigned long long)st.dr_n[c_]); } } } wt{(unsigned long long)wall_clock64(), wt_tag, wst};
    StreamStats *const wss = wt_tag == 0 ? &wst : nullptr;
#else
    StreamStats *const wss = nullptr;
#endif
    const bool strided = (even_waves >> 31) != 0u; even_waves &= 0x7FFFFFFFu;
    if (even_waves) rays_per_wave = max(64u, ((n + even_waves - 1u) / even_waves + 63u) & ~63u);
    const uint32_t begin = blockIdx.x * rays_per_wave;
    if (begin >= n) return;
    StreamExt<true> ext{reinterpret_cast<float *>(lds_dyn), lds_dyn + 256u};
    // stream_stride: the (n + rays_per_wave - 1) / rays_per_wave waves that have work take the queue's 64-ray batches round-robin instead of one contiguous range each
    const BatchStride src = strided ? BatchStride{blockIdx.x * 64u, 64u * ((n + rays_per_wave - 1u) / rays_per_wave), n} : BatchStride{begin, 64u, min(n, begin + rays_per_wave)};
    traverse_wide_stream<false, false, false, NoPairs, StreamExt<true>>(s, src, lds_dyn + HIT_LDS_WORDS,
        [&](uint32_t i, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) {
            const bool sh = i >= n_next; tag = sh ? i - n_next : i; is_any = sh ? 1u : 0u;
            A = qload(sh ? &srayA[tag] : &rayA[tag]); B = qload(sh ? &srayB[tag] : &rayB[tag]);
            if (!sh) A.w = __builtin_inff();          // a bounce ray's tmax word may carry the throughput chain
            else tag = __float_as_uint(B.w);          // shadow planes: the ray reports to its pixel's byte
        },
        [&](uint32_t j, bool is_any, bool hit, const TravHit &h) {
            if (is_any) { if (!hit) lit[4 * (size_t)j] = 1; }
            else qstore(&hits[j], hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)));
        }, wss, NoPairs{}, ext);
}

#include "megakernel.h"            // k_megakernel: one launch per frame

// Primary rays on the wide layout with lane refill (experiment: the rope kernel is VALU-bound on primary rays; measured
// equal on the full frame, 7 % slower on the primary + shadow workload).
template <bool TWO_LEVEL, bool SEED>
__global__ void __launch_bounds__(64, 5) k_trace_primary_wide_stream(SceneView s, FrameParams fp, const uint32_t *__restrict__ seeds, float4 *__restrict__ hits, float4 *__restrict__ dirs, uint32_t capacity, uint32_t rays_per_wave,
                                                                     uint32_t *__restrict__ hint /* SEED: per pixel, the (packet | instance << 24) its primary ray hit last */) {
    extern __shared__ uint32_t stk_dyn[];
    const uint32_t begin = blockIdx.x * rays_per_wave, sub = blockIdx.y;
    if (begin >= capacity) return;
    hits += (size_t)sub * capacity; dirs += (size_t)sub * capacity;
    auto make_ray = [&](uint32_t slot, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any, f3 &org, f3 &dir, int &x, int &y) -> bool {
        is_any = 0u; tag = slot;
        if (slot_to_pixel(fp, slot, x, y)) {
            primary_ray(fp, seeds, sub * fp.capacity + slot, x, y, org, dir);
            A = make_float4(org.x, org.y, org.z, __builtin_inff()); B = make_float4(dir.x, dir.y, dir.z, 0.0f);
            dirs[slot] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(sub * fp.capacity + slot));   // read back by k_shade
            return true;
        }
        A = make_float4(0.0f, 0.0f, 0.0f, -1.0f); B = make_float4(0.0f, 0.0f, 1.0f, 0.0f);                // partial-tile slot: tmax < 0 -> miss
        return false;
    };
    if constexpr (SEED) {
        // the triangle this pixel hit in an earlier frame is tested first (in its instance's object space: the arithmetic of the walk itself): the jittered ray most
        // often hits it again and the walk starts with the right distance bound.  A guess is legal when its packet belongs to its instance's BLAS (a stale one —
        // another scene, moved instances — is then one wasted test or none); the result is the minimum over (t, id) either way.
        traverse_wide_stream<TWO_LEVEL, true>(s, OneRange{begin, min(capacity, begin + rays_per_wave)}, stk_dyn,
            [&](uint32_t slot, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any, uint32_t &seed) {
                f3 org, dir; int x, y;
                seed = 0xFFFFFFFFu;
                if (!make_ray(slot, A, B, tag, is_any, org, dir, x, y)) return;
                const uint32_t guess = hint[(uint32_t)y * (uint32_t)fp.width + (uint32_t)x];
                const uint32_t pk = TWO_LEVEL ? (guess & 0xFFFFFFu) : guess, in = TWO_LEVEL ? guess >> 24 : 0u;
                if (guess == 0xFFFFFFFFu) return;
                float t, U, V, ad;
                if (TWO_LEVEL) {
                    if (in >= s.num_inst) return;
                    const InstanceDev &I = s.inst[in];
                    if (pk - I.packet_base >= I.ntri) return;
                    const float4 *__restrict__ q = s.wpackets + WPK * (size_t)pk;
                    if (tri_test(q[0], q[1], q[2], to_object_point(I, org), to_object_dir(I, dir), 0.0f, __builtin_inff(), t, U, V, ad)) { A.w = t; seed = guess; }
                } else {
                    if (pk >= s.num_tris) return;
                    const float4 *__restrict__ q = s.wpackets + WPK * (size_t)pk;
                    if (tri_test(q[0], q[1], q[2], org, dir, 0.0f, __builtin_inff(), t, U, V, ad)) { A.w = t; seed = guess; }
                }
            },
            [&](uint32_t slot, bool, bool hit, const TravHit &h) {
                hits[slot] = hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                int x, y;
                if (hit && slot_to_pixel(fp, slot, x, y)) hint[(uint32_t)y * (uint32_t)fp.width + (uint32_t)x] = h.pk;
            });
    } else {
        traverse_wide_stream<TWO_LEVEL>(s, OneRange{begin, min(capacity, begin + rays_per_wave)}, stk_dyn,
            [&](uint32_t slot, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) { f3 org, dir; int x, y; (void)make_ray(slot, A, B, tag, is_any, org, dir, x, y); },
            [&](uint32_t slot, bool, bool hit, const TravHit &h) {
                hits[slot] = hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
            });
    }
}

// ------------------------------------------------------------------ accumulate (Raytracing.metal:394-403)
// Also the frame's bookkeeping (block 0, thread 0): per-bounce queue counters {next rays, shadow rays} are
// folded into the running totals and zeroed for the next frame.
constexpr size_t WORK_COUNTERS = 128, WORK_COUNTERS_PER_BOUNCE = 8 * XCD_COUNTER_STRIDE / 2;      // in 64-bit words of FrameLane::bounce_counts
__global__ void __launch_bounds__(64) k_accumulate(FrameParams fp, const float4 *__restrict__ sample, const float4 *__restrict__ prev, float4 *__restrict__ dst,
                                                   unsigned long long *__restrict__ bounce_counts, unsigned long long *__restrict__ totals, uint32_t primary) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long closest = primary, shadow = 0;
        for (int b = 0; b < fp.max_bounces; b++) {
            unsigned long long c = bounce_counts[b];
            if (b + 1 < fp.max_bounces) closest += (uint32_t)c;
            shadow += c >> 32;
            bounce_counts[b] = 0;
            bounce_counts[32 + b] = 0;               // work counter of the TLAS pass of this bounce (two-level scenes, binned)
            for (int x = 0; x < 8; x++) reinterpret_cast<uint32_t *>(bounce_counts + WORK_COUNTERS + (size_t)b * WORK_COUNTERS_PER_BOUNCE)[x * XCD_COUNTER_STRIDE] = 0;      // work counters of the persistent trace launch of this bounce (one per XCD region)
            bounce_counts[65 + b] = 0;               // two-level scenes, binned: {pairs queued (lo), work counter of the BLAS pass (hi)}
        }
        atomicAdd(&totals[0], closest); atomicAdd(&totals[1], shadow); atomicAdd(&totals[2], (unsigned long long)primary);      // (the tile groups of a pass accumulate side by side)
    }
    uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    int x, y;
    if (!slot_to_pixel(fp, slot, x, y)) return;
    uint32_t pix = (uint32_t)y * (uint32_t)fp.width + (uint32_t)x;
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int sub = 0; sub < fp.batch; sub++) {                           // the batch's frames, in frame order
        const float4 sm = qload(&sample[(size_t)sub * fp.capacity + slot]);
        const uint32_t frame = fp.frameIndex + (uint32_t)sub;
        if (frame > 0) {
            const float4 p = sub == 0 ? q2load(&prev[pix]) : c;
            float fi = (float)frame;
            float den = (float)(frame + 1);
            c.x = (sm.x + p.x * fi) / den; c.y = (sm.y + p.y * fi) / den; c.z = (sm.z + p.z * fi) / den;
        } else c = sm;
    }
    q2store(&dst[pix], make_float4(c.x, c.y, c.z, 1.0f));
}

// Shadow planes (default): shade(b) leaves the light's contribution in con[b][pixel], a shadow ray that gets
// through sets lit[pixel][b] (one byte of the four a pixel has per frame), and the pixel's sample is the sum of the contributions whose byte is set, in bounce order — the additions of
// Raytracing.metal:371-373 on the same floats in the same order as the read-modify-write of one sample buffer made them (0 + c0, + c1, + c2).  Per shadow ray
// that is 16 bytes written and one byte instead of 32 bytes through the queue and a 32-byte read-modify-write inside the traversal loop.
MRT_DEV float4 planes_sample(const float4 *const (&con)[3], const uint8_t *__restrict__ lit, int max_bounces, size_t sp) {
    float4 sm = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t flags = reinterpret_cast<const uint32_t *>(lit)[sp];      // byte b: the shadow ray of bounce b got through
#pragma unroll
    for (int b = 0; b < 3; b++)
        if (b < max_bounces && ((flags >> (8 * b)) & 0xFFu) != 0u) { const float4 cc = qload(&con[b][sp]); sm = make_float4(sm.x + cc.x, sm.y + cc.y, sm.z + cc.z, 0.0f); }
    return sm;
}
__global__ void __launch_bounds__(64) k_accumulate_planes(FrameParams fp, const float4 *__restrict__ con0, const float4 *__restrict__ con1, const float4 *__restrict__ con2, const uint8_t *__restrict__ lit,
                                                          const float4 *__restrict__ prev, float4 *__restrict__ dst, unsigned long long *__restrict__ bounce_counts, unsigned long long *__restrict__ totals, uint32_t primary) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long closest = primary, shadow = 0;
        for (int b = 0; b < fp.max_bounces; b++) {
            unsigned long long c = bounce_counts[b];
            if (b + 1 < fp.max_bounces) closest += (uint32_t)c;
            shadow += c >> 32;
            bounce_counts[b] = 0;
            bounce_counts[32 + b] = 0;               // work counter of the TLAS pass of this bounce (two-level scenes, binned)
            for (int x = 0; x < 8; x++) reinterpret_cast<uint32_t *>(bounce_counts + WORK_COUNTERS + (size_t)b * WORK_COUNTERS_PER_BOUNCE)[x * XCD_COUNTER_STRIDE] = 0;      // work counters of the persistent trace launch of this bounce (one per XCD region)
            bounce_counts[65 + b] = 0;               // two-level scenes, binned: {pairs queued (lo), work counter of the BLAS pass (hi)}
        }
        atomicAdd(&totals[0], closest); atomicAdd(&totals[1], shadow); atomicAdd(&totals[2], (unsigned long long)primary);
    }
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    int x, y;
    if (!slot_to_pixel(fp, slot, x, y)) return;
    const uint32_t pix = (uint32_t)y * (uint32_t)fp.width + (uint32_t)x;
    const float4 *const con[3] = {con0, con1, con2};
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int sub = 0; sub < fp.batch; sub++) {                           // the batch's frames, in frame order
        const float4 sm = planes_sample(con, lit, fp.max_bounces, (size_t)sub * fp.capacity + slot);
        const uint32_t frame = fp.frameIndex + (uint32_t)sub;
        if (frame > 0) {
            const float4 p = sub == 0 ? q2load(&prev[pix]) : c;
            const float fi = (float)frame, den = (float)(frame + 1);
            c.x = (sm.x + p.x * fi) / den; c.y = (sm.y + p.y * fi) / den; c.z = (sm.z + p.z * fi) / den;
        } else c = sm;
    }
    q2store(&dst[pix], make_float4(c.x, c.y, c.z, 1.0f));
}

// The last passes of a draw, accumulated in ONE launch on the main stream (Renderer::tail_accumulate): their accumulate launches would otherwise run one after the other at the very
// end of the call — each reads the previous one's output — when nothing else is left to overlap them with (rocprofv3, the driver's 20 steps: five launches of 60-90 us plus their
// gaps, 0.6 of 12.8 ms).  Same folds in the same order (pass by pass, frame by frame); the accumulation target is read once and written once.
struct AccPass { const float4 *con0, *con1, *con2; const uint8_t *lit; unsigned long long *counts; uint32_t frameIndex; int32_t batch; uint32_t primary; uint32_t pad; };
struct AccGroup { AccPass p[MAX_FRAMES_IN_FLIGHT]; int32_t n; };
__global__ void __launch_bounds__(64) k_accumulate_planes_group(FrameParams fp, AccGroup g, const float4 *__restrict__ prev, float4 *__restrict__ dst, unsigned long long *__restrict__ totals) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long closest = 0, shadow = 0, primary = 0;
        for (int p = 0; p < MAX_FRAMES_IN_FLIGHT; p++) {
            if (p >= g.n) break;
            closest += g.p[p].primary; primary += g.p[p].primary;
            for (int b = 0; b < fp.max_bounces; b++) {
                const unsigned long long c = g.p[p].counts[b];
                if (b + 1 < fp.max_bounces) closest += (uint32_t)c;
                shadow += c >> 32;
                g.p[p].counts[b] = 0; g.p[p].counts[32 + b] = 0; g.p[p].counts[65 + b] = 0;
                for (int x = 0; x < 8; x++) reinterpret_cast<uint32_t *>(g.p[p].counts + WORK_COUNTERS + (size_t)b * WORK_COUNTERS_PER_BOUNCE)[x * XCD_COUNTER_STRIDE] = 0;
            }
        }
        totals[0] += closest; totals[1] += shadow; totals[2] += primary;
    }
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    int x, y;
    if (!slot_to_pixel(fp, slot, x, y)) return;
    const uint32_t pix = (uint32_t)y * (uint32_t)fp.width + (uint32_t)x;
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool first = true;
    for (int p = 0; p < MAX_FRAMES_IN_FLIGHT; p++) {
        if (p >= g.n) break;
        const float4 *const con[3] = {g.p[p].con0, g.p[p].con1, g.p[p].con2};
        for (int sub = 0; sub < g.p[p].batch; sub++) {
            const float4 sm = planes_sample(con, g.p[p].lit, fp.max_bounces, (size_t)sub * fp.capacity + slot);
            const uint32_t frame = g.p[p].frameIndex + (uint32_t)sub;
            if (frame > 0) {
                const float4 q = first ? q2load(&prev[pix]) : c;
                const float fi = (float)frame, den = (float)(frame + 1);
                c.x = (sm.x + q.x * fi) / den; c.y = (sm.y + q.y * fi) / den; c.z = (sm.z + q.z * fi) / den;
            } else c = sm;
            first = false;
        }
    }
    q2store(&dst[pix], make_float4(c.x, c.y, c.z, 1.0f));
}

// A shard's own pixels as a compact buffer, and back (the compact assemble of a tile-sharded image: mrt_renderer_pack_owned_tiles / _unpack_tiles, mrt_group reduce mode 2).
// Layout: tiles_local x 64 RGBA32F — tile lt of shard (rank, world) is tile lt * world + rank of the image, its 8 x 8 pixels row-major; pixels of an edge tile
// outside the image are 0 in the buffer and skipped on the way back.  1 / world of the image instead of the full frame the reduce(sum) moves per rank.
template <bool PACK>
__global__ void k_tiles(float4 *__restrict__ image, int w, int h, int tiles_x, int rank, int world, uint32_t tiles_local, float4 *__restrict__ compact) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x, lt = slot >> 6, k = slot & 63u;
    if (lt >= tiles_local) return;
    const uint32_t tile = lt * (uint32_t)world + (uint32_t)rank, ty = tile / (uint32_t)tiles_x, tx = tile - ty * (uint32_t)tiles_x;
    const int x = (int)(tx * 8u + (k & 7u)), y = (int)(ty * 8u + (k >> 3));
    const bool in = x < w && y < h;
    if (PACK) compact[slot] = in ? image[(size_t)y * w + x] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    else if (in) image[(size_t)y * w + x] = compact[slot];
}

// Shaders.metal:39-52 — Reinhard + vertical flip (the blit's uv, :35), RGBA8
__global__ void k_tonemap(const float4 *__restrict__ accum, int w, int h, uchar4 *__restrict__ out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float4 a = accum[(size_t)(h - 1 - y) * w + x];
    float c[3] = {a.x, a.y, a.z};
    unsigned char o[3];
    for (int k = 0; k < 3; k++) { float v = c[k] / (1.0f + c[k]); v = saturatef(v); o[k] = (unsigned char)(v * 255.0f + 0.5f); }
    out[(size_t)y * w + x] = make_uchar4(o[0], o[1], o[2], 255);
}

// ------------------------------------------------------------------ query kernels (C-ABI intersect_*)
template <bool WIDE>
__global__ void __launch_bounds__(64) k_query_closest(SceneView s, const MRTRay *__restrict__ rays, uint32_t n, MRTIntersection *__restrict__ out) {
    extern __shared__ uint32_t stk[];      // WIDE: the scene's wide-tree depth x WIDE_STACK_LEVEL_BYTES, sized by the host
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    MRTRay r = rays[i];
    TravHit h;
    const f3 ro = mk3(r.origin[0], r.origin[1], r.origin[2]), rd = mk3(r.direction[0], r.direction[1], r.direction[2]);
    bool hit = s.num_inst ? traverse_instanced<false>(s, ro, rd, r.min_distance, r.max_distance, h)
             : WIDE ? traverse_wide<false>(s, ro, rd, r.min_distance, r.max_distance, h, stk) : traverse<false>(s, ro, rd, r.min_distance, r.max_distance, h);
    MRTIntersection o;
    o._pad = 0;
    if (hit) {
        uint32_t inst, geom;
        if (s.num_inst) { inst = instance_of_gid(s, h.gid); const InstanceDev &I = s.inst[inst]; geom = s.tri_shade[I.ts_base + (h.gid - I.gid_base)].w & 0xFFFFu; }
        else { const uint4 ts = s.tri_shade[h.gid]; inst = ts.w >> 16; geom = ts.w & 0xFFFFu; }
        o.type = 1; o.distance = h.t; o.instance_id = (int32_t)inst; o.geometry_id = (int32_t)geom;
        o.primitive_id = (int32_t)(h.gid - s.geom_base[inst * (uint32_t)s.max_sub + geom]);
        o.u = h.U / h.ad; o.v = h.V / h.ad;
    } else { o.type = 0; o.distance = -1.0f; o.instance_id = o.geometry_id = o.primitive_id = -1; o.u = o.v = 0.0f; }
    out[i] = o;
}
template <bool WIDE>
__global__ void __launch_bounds__(64) k_query_any(SceneView s, const MRTRay *__restrict__ rays, uint32_t n, int32_t *__restrict__ out) {
    extern __shared__ uint32_t stk[];      // WIDE: the scene's wide-tree depth x WIDE_STACK_LEVEL_BYTES, sized by the host
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    MRTRay r = rays[i];
    TravHit h;
    const f3 ro = mk3(r.origin[0], r.origin[1], r.origin[2]), rd = mk3(r.direction[0], r.direction[1], r.direction[2]);
    out[i] = (s.num_inst ? traverse_instanced<true>(s, ro, rd, r.min_distance, r.max_distance, h)
              : WIDE ? traverse_wide<true>(s, ro, rd, r.min_distance, r.max_distance, h, stk) : traverse<true>(s, ro, rd, r.min_distance, r.max_distance, h)) ? 1 : 0;
}

// per-ray traversal statistics (steps, leaf visits, triangle tests) — diagnostics only
template <bool WIDE>
__global__ void __launch_bounds__(64) k_query_stats(SceneView s, const MRTRay *__restrict__ rays, uint32_t n, int any, uint32_t *__restrict__ out) {
    extern __shared__ uint32_t stk[];      // WIDE: the scene's wide-tree depth x WIDE_STACK_LEVEL_BYTES, sized by the host
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    MRTRay r = rays[i];
    TravHit h; TravCounters tc{0, 0, 0, 0};
    tc.alu_dup = (any >> 8) & 0xFF; tc.mem_dup = (any >> 16) & 0xFF; any &= 1;
    f3 o = mk3(r.origin[0], r.origin[1], r.origin[2]), d = mk3(r.direction[0], r.direction[1], r.direction[2]);
    unsigned long long t0 = wall_clock64();
    if (WIDE) { if (any) traverse_wide<true, true>(s, o, d, r.min_distance, r.max_distance, h, stk, &tc); else traverse_wide<false, true>(s, o, d, r.min_distance, r.max_distance, h, stk, &tc); }
    else { if (any) traverse<true, true>(s, o, d, r.min_distance, r.max_distance, h, &tc); else traverse<false, true>(s, o, d, r.min_distance, r.max_distance, h, &tc); }
    unsigned long long t1 = wall_clock64();
    out[8 * i + 0] = tc.steps; out[8 * i + 1] = tc.leaves; out[8 * i + 2] = tc.tris; out[8 * i + 3] = h.gid;
    out[8 * i + 4] = (uint32_t)t0; out[8 * i + 5] = (uint32_t)t1; out[8 * i + 6] = tc.wave_iters;   // 100 MHz ticks
    out[8 * i + 7] = (tc.alu_dup | tc.mem_dup) ? __float_as_uint(tc.sink) : (tc.empty & 0xFFFFu) | (tc.stale << 16);    // 8-wide layout: empty visits | stale visits << 16
}

// stream-traversal lane accounting (diagnostics): per wave {iterations, sum of live lanes, node lanes, tri lanes, refills, refilled lanes}
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(64) k_query_stream_stats(SceneView s, const MRTRay *__restrict__ rays, uint32_t n, int any, uint32_t per_wave, uint32_t depth, uint32_t *__restrict__ out) {
    extern __shared__ uint32_t stk_dyn[];
    const uint32_t begin = blockIdx.x * per_wave;
    if (begin >= n) return;
    const uint32_t end = min(n, begin + per_wave);
    StreamStats ss{0, 0, 0, 0, 0, 0};
    uint32_t sink = 0;
#if defined(MRT_STATS_PROBE) && MRT_STATS_PROBE == 1
    {   // BFS numbering, children contiguous: level L + 1 starts where level L ends and holds the internal children (imask bits) of level L's nodes
        uint32_t lo = 0, hi = 1;
        for (int L = 0; L < 3; L++) {
            uint32_t kids = 0;
            for (uint32_t i = lo; i < hi; i++) kids += (uint32_t)__popc(__float_as_uint(s.wnodes[WNODE_STRIDE * (size_t)i].w) >> 24);
            lo = hi; hi += kids; ss.level_end[L] = hi;
        }
    }
#endif
    traverse_wide_stream<TWO_LEVEL>(s, OneRange{begin, end}, stk_dyn,
        [&](uint32_t i, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) {
            MRTRay r = rays[i]; tag = i & 0x7FFFFFFFu;
            A = make_float4(r.origin[0], r.origin[1], r.origin[2], r.max_distance); B = make_float4(r.direction[0], r.direction[1], r.direction[2], 0.0f); is_any = (uint32_t)any;
        },
        [&](uint32_t, bool, bool hit, const TravHit &) { sink += hit ? 1u : 0u; }, &ss);
    if ((threadIdx.x & 63) == 0) {
        uint32_t *o = out + 8 * (size_t)blockIdx.x;
        o[0] = ss.iters; o[1] = ss.live_sum; o[2] = ss.node_sum; o[3] = ss.tri_sum; o[4] = ss.refills; o[5] = ss.refill_lanes; o[6] = sink; o[7] = end - begin;
#ifdef MRT_STATS_PROBE
        o[4] = ss.probe[0]; o[5] = ss.probe[1]; o[6] = ss.probe[2];
#endif
    }
}

// the render kernels' traversal (8-wide stream with lane refill; both levels of an instanced scene) on caller rays — parity tests of that path
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(64) k_query_stream(SceneView s, const MRTRay *__restrict__ rays, uint32_t n, int any, uint32_t per_wave, MRTIntersection *__restrict__ out) {
    extern __shared__ uint32_t stk_dyn[];
    const uint32_t begin = blockIdx.x * per_wave;
    if (begin >= n) return;
    traverse_wide_stream<TWO_LEVEL>(s, OneRange{begin, min(n, begin + per_wave)}, stk_dyn,
        [&](uint32_t i, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) {
            MRTRay r = rays[i]; tag = i & 0x7FFFFFFFu; is_any = (uint32_t)any;
            A = make_float4(r.origin[0], r.origin[1], r.origin[2], r.max_distance); B = make_float4(r.direction[0], r.direction[1], r.direction[2], 0.0f);
        },
        [&](uint32_t i, bool is_any, bool hit, const TravHit &h) {
            MRTIntersection o;
            o._pad = 0; o.type = hit ? 1 : 0; o.distance = -1.0f; o.instance_id = o.geometry_id = o.primitive_id = -1; o.u = o.v = 0.0f;
            if (hit && !is_any) {
                uint32_t inst, geom;
                if (TWO_LEVEL) { inst = instance_of_gid(s, h.gid); const InstanceDev &I = s.inst[inst]; geom = s.tri_shade[I.ts_base + (h.gid - I.gid_base)].w & 0xFFFFu; }
                else { const uint4 ts = s.tri_shade[h.gid]; inst = ts.w >> 16; geom = ts.w & 0xFFFFu; }
                o.distance = h.t; o.instance_id = (int32_t)inst; o.geometry_id = (int32_t)geom;
                o.primitive_id = (int32_t)(h.gid - s.geom_base[inst * (uint32_t)s.max_sub + geom]);
                o.u = h.U / h.ad; o.v = h.V / h.ad;
            }
            out[i] = o;
        });
}

// ------------------------------------------------------------------ device-function probes
__global__ void k_probe_halton(const int32_t *i, const int32_t *d, uint32_t n, float *out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = halton_dev(i[k], d[k]);
}
__global__ void k_probe_hemisphere(const float *u2, const float *n3, uint32_t n, float *out3) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    f3 v = align_hemisphere_dev(sample_cosine_hemisphere_dev(u2[2 * k], u2[2 * k + 1]), mk3(n3[3 * k], n3[3 * k + 1], n3[3 * k + 2]));
    out3[3 * k] = v.x; out3[3 * k + 1] = v.y; out3[3 * k + 2] = v.z;
}

static inline uint32_t cdiv(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

// Launch with the kernel's own start/stop events when `ev` is given (bench.py's live roofline measurement).
template <class... KArgs, class... Args>
static inline void launch_timed(EvPair *ev, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t st, Args... args) {
    if (ev) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)shmem, st, ev->a, ev->b, 0, static_cast<KArgs>(args)...);
    else hipLaunchKernelGGL(kernel, grid, block, shmem, st, static_cast<KArgs>(args)...);
}

}  // namespace

#ifdef MRT_WAVE_TIMES
int read_wave_times(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), sizeof(g_wave_times)) == hipSuccess ? MRT_OK : MRT_ERR_HIP; }
int read_wave_iters(uint32_t *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_iters), sizeof(g_wave_iters)) == hipSuccess ? MRT_OK : MRT_ERR_HIP; }
int read_drain_probe(unsigned long long *out18, int reset) {
    if (hipMemcpyFromSymbol(out18, HIP_SYMBOL(g_drain_probe), sizeof(g_drain_probe)) != hipSuccess) return MRT_ERR_HIP;
    if (reset) { static const unsigned long long z[18] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_drain_probe), z, sizeof z) != hipSuccess) return MRT_ERR_HIP; }
    return MRT_OK;
}
#endif

// ====================================================================== Renderer (host)
// The table of bounce 0's Halton values (FrameParams::halton_tab) for the indices [w0, w0 + HALTON_TAB_SPAN): a pass needs offset + sampleIndex + sub-frame for offsets below 2^20.
// Once per draw, on the MAIN stream ahead of the fork: every lane of this draw starts behind it (the fork event), and every pass of earlier draws has been joined into the main stream
// — no host or device synchronisation (until round 6 a window move drained the whole device from inside the enqueue loop).  The window follows the draw's first frame when the draw's
// first pass would leave it; what a very long draw reaches beyond the window falls back to the recurrence per index (halton_b0), bit-identical either way.
int Renderer::ensure_halton_table(uint32_t sample_index, uint32_t frames) {
    const uint64_t first_pass = std::min<uint64_t>(frames, MAX_FRAME_BATCH);
    const bool fits = halton_tab.p != nullptr && sample_index >= halton_w0 && (uint64_t)sample_index + first_pass + (1u << 20) <= (uint64_t)halton_w0 + HALTON_TAB_SPAN;
    if (fits) return MRT_OK;
    if (!halton_tab.p) MRT_HIP(halton_tab.alloc((size_t)HALTON_TAB_DIMS * HALTON_TAB_SPAN));
    halton_w0 = sample_index;
    hipLaunchKernelGGL(k_halton_table, dim3(cdiv(HALTON_TAB_SPAN, 256), HALTON_TAB_DIMS), dim3(256), 0, stream, halton_tab.p, halton_w0, HALTON_TAB_SPAN);
    MRT_HIP(hipGetLastError());
    return MRT_OK;
}

int Renderer::init(hipStream_t st, const DeviceScene *sc, int w, int h, uint32_t seed_, int max_bounces_) {
    stream = st; scene = sc; seed = seed_; max_bounces = max_bounces_;
    MRT_HIP(hipEventCreate(&ev_begin)); MRT_HIP(hipEventCreate(&ev_end));
    MRT_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    for (auto &e : ev_ext) { MRT_HIP(hipEventCreate(&e.a)); MRT_HIP(hipEventCreate(&e.b)); }
    for (auto &L : lanes) {
        MRT_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        MRT_HIP(hipEventCreateWithFlags(&L.accumulated, hipEventDisableTiming));
        // per bounce b: [b] queue counts {next rays (lo 32), shadow rays (hi 32)}; [32 + b] work counter of the TLAS pass and [65 + b] {pairs queued, work counter of the BLAS pass} (two-level scenes, binned);
        // from WORK_COUNTERS on: the eight work counters of the pulling traversal launch, 128 bytes apart (XcdRegions)
        MRT_HIP(L.bounce_counts.alloc(WORK_COUNTERS + 32 * WORK_COUNTERS_PER_BOUNCE));
        MRT_HIP(hipMemsetAsync(L.bounce_counts.p, 0, L.bounce_counts.bytes(), stream));
    }
    MRT_HIP(totals.alloc(4));
    MRT_HIP(hipMemsetAsync(totals.p, 0, totals.bytes(), stream));
    return resize(w, h);
}

Renderer::~Renderer() {
    if (ev_begin) (void)hipEventDestroy(ev_begin);
    if (ev_end) (void)hipEventDestroy(ev_end);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (auto &e : ev_ext) { if (e.a) (void)hipEventDestroy(e.a); if (e.b) (void)hipEventDestroy(e.b); }
    for (auto &pd : passes_pending) (void)hipEventDestroy(pd.ev);
    for (auto e : pass_events_free) (void)hipEventDestroy(e);
    for (auto &L : lanes) {
        if (L.stream) { (void)hipStreamSynchronize(L.stream); (void)hipStreamDestroy(L.stream); }
        if (L.accumulated) (void)hipEventDestroy(L.accumulated);
    }
}

// frame_batch = 0: a pass carries about as many pixel-frames as eight 1080p frames, whatever the image — a launch over 256 x 256 pixels x 8 frames is a sixteenth of one the kernels were
// tuned on and the pipeline is launch-bound there (Cornell 256^2: 6.7 Grays/s in 8-frame passes, 9.8-10.8 in 32-frame ones; 640 x 360: 9.0 -> 10.4; 960 x 540: 10.9 -> 11.5).  The rule a
// tile-sharded rank follows (distributed.py shard_frame_batch: 8 x world) is the same rule: its pixels are a world-th of the image.
int Renderer::batch_wanted() const {
    if (frame_batch > 0) return std::min(frame_batch, MAX_FRAME_BATCH);
    const size_t owned = std::max<size_t>(1, (size_t)std::max(width, 1) * (size_t)std::max(height, 1) / (size_t)std::max(shard_world, 1));
    const size_t ref = (size_t)DEFAULT_FRAME_BATCH * 1920 * 1080;
    return (int)std::min<size_t>(MAX_FRAME_BATCH, std::max<size_t>(DEFAULT_FRAME_BATCH, (ref + owned - 1) / owned));
}

int Renderer::resize(int w, int h) {                                   // Renderer.swift:353-356 → createTextures :231-275
    width = w; height = h;
    const size_t npix = (size_t)w * h;
    const int B = batch_wanted();
    alloc_batch = B;
    MRT_HIP(hint.alloc(npix));
    MRT_HIP(hipMemsetAsync(hint.p, 0xFF, hint.bytes(), stream));
    MRT_HIP(accum[0].alloc(npix)); MRT_HIP(accum[1].alloc(npix));
    MRT_HIP(hipMemsetAsync(accum[0].p, 0, accum[0].bytes(), stream));
    MRT_HIP(hipMemsetAsync(accum[1].p, 0, accum[1].bytes(), stream));
    default_camera(w, h, &camera);
    frame_index = 0; cur = 0;
    MRT_HIP(hipMemsetAsync(totals.p, 0, totals.bytes(), stream));
    frames_rendered = 0;
    for (auto &pd : passes_pending) pass_events_free.push_back(pd.ev);      // the caller synchronised (mrt_renderer_resize)
    passes_pending.clear(); frames_completed_known = 0;
    return alloc_queues();
}

int Renderer::note_pass(hipStream_t st) {
    // more than 1024 passes queued ahead of the GPU: the newest entry absorbs further passes (its event is re-recorded)
    if (passes_pending.size() < 1024 || passes_pending.empty()) {
        hipEvent_t e = nullptr;
        if (!pass_events_free.empty()) { e = pass_events_free.back(); pass_events_free.pop_back(); }
        else MRT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        passes_pending.push_back(PassDone{e, frames_rendered});
    } else passes_pending.back().frames_through = frames_rendered;
    MRT_HIP(hipEventRecord(passes_pending.back().ev, st));
    return MRT_OK;
}
int Renderer::poll_completed(uint64_t *out) {
    while (!passes_pending.empty()) {
        const hipError_t q = hipEventQuery(passes_pending.front().ev);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }
        if (q != hipSuccess) return hip_fail(q, "hipEventQuery", __FILE__, __LINE__);
        frames_completed_known = passes_pending.front().frames_through;
        pass_events_free.push_back(passes_pending.front().ev);
        passes_pending.pop_front();
    }
    *out = frames_completed_known;
    return MRT_OK;
}

int Renderer::alloc_queues() {
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    const int tiles = tiles_x * tiles_y;
    tiles_local = (tiles - shard_rank + shard_world - 1) / shard_world;
    if (tiles_local < 0) tiles_local = 0;
    capacity = (uint32_t)tiles_local * 64u;
    {   // the seed table, by sample index (sub-frame * capacity + slot): this shard's pixels only
        const int B = std::max(1, alloc_batch);
        MRT_HIP(seeds.alloc((size_t)std::max<uint32_t>(capacity, 1u) * B));
        FrameParams sp{}; sp.width = width; sp.height = height; sp.shard_rank = shard_rank; sp.shard_world = shard_world; sp.tiles_x = tiles_x; sp.tiles_local = tiles_local; sp.capacity = capacity;
        if (capacity) hipLaunchKernelGGL(k_seed_slots, dim3(cdiv(capacity, 256), B), dim3(256), 0, stream, seeds.p, sp, seed);
    }
    tgroups_for = 0;            // the tile groups' seed tables follow the shard and the image size: rebuilt at the next draw that uses groups
    // lanes get their buffers when first used (alloc_lane): 16 lanes x 4-frame queues would pin 23 GB at 1080p, 94 GB at 4K
    lanes_ready = 0;
    for (auto &L : lanes) release_lane(L);
    // pixels owned by this shard (edge tiles may be partial)
    uint64_t owned = 0;
    for (int lt = 0; lt < tiles_local; lt++) {
        int tile = lt * shard_world + shard_rank, ty = tile / tiles_x, tx = tile % tiles_x;
        int pw = std::min(8, width - tx * 8), ph = std::min(8, height - ty * 8);
        owned += (uint64_t)pw * ph;
    }
    owned_pixels = owned;
    return MRT_OK;
}

constexpr int PLANES_MAX_BOUNCES = 3;        // shadow planes: sample + two more contribution planes, three flag bytes of the pixel's word

size_t Renderer::lane_bytes() const {
    const size_t qcap = (size_t)capacity * (size_t)std::max(1, alloc_batch);
    const bool need_thr = !(throughput_chain && !materials && max_bounces <= 3);
    const size_t spix = (size_t)std::max<uint32_t>(capacity, 1u) * (size_t)std::max(1, alloc_batch);      // sample indices of a pass: sub-frame * capacity + slot
    const size_t planes_bytes = shadow_planes ? 2 * spix * sizeof(float4) + spix * 4 : 0;
    const bool need_scon = need_thr || !shadow_planes;
    const size_t pairs_bytes = (scene && scene->num_inst && scene->num_wnodes && tl_pairs && shadow_planes && !need_thr) ? 2 * PairQueue::WORDS * qcap * sizeof(uint4) : 0;      // two-level scenes: the (ray, instance) queue of the binned walk
    return ((need_thr ? 9 : 7) * qcap + (need_scon ? qcap : 0) + spix) * sizeof(float4) + planes_bytes + pairs_bytes;
}
void Renderer::release_lane(FrameLane &L) {
    for (int k = 0; k < 2; k++) { L.rayA[k].release(); L.rayB[k].release(); L.thr[k].release(); L.f_con[k].release(); }
    L.hits.release(); L.srayA.release(); L.srayB.release(); L.scon.release(); L.sample.release(); L.f_lit.release(); L.pairs.release();
}
int Renderer::alloc_planes(FrameLane &L) {
    const size_t spix = (size_t)std::max<uint32_t>(capacity, 1u) * (size_t)std::max(1, alloc_batch);      // sample indices of a pass: sub-frame * capacity + slot
    for (int k = 0; k < 2; k++) MRT_HIP(L.f_con[k].alloc(spix));
    MRT_HIP(L.f_lit.alloc(spix * 4));          // [sub-frame][pixel][bounce]: one 32-bit word per pixel and frame
    return MRT_OK;
}
int Renderer::alloc_lane(FrameLane &L) {
    const size_t qcap = (size_t)capacity * (size_t)std::max(1, alloc_batch);      // a batch of frames shares one set of queues
    const bool need_thr = !(throughput_chain && !materials && max_bounces <= 3);      // else on demand (render())
    for (int k = 0; k < 2; k++) { MRT_HIP(L.rayA[k].alloc(qcap)); MRT_HIP(L.rayB[k].alloc(qcap)); if (need_thr) MRT_HIP(L.thr[k].alloc(qcap)); }
    MRT_HIP(L.hits.alloc(qcap)); MRT_HIP(L.srayA.alloc(qcap)); MRT_HIP(L.srayB.alloc(qcap));
    if (need_thr || !shadow_planes) MRT_HIP(L.scon.alloc(qcap));          // the contribution queue: with shadow planes only the passes they do not cover need it (allocated then, render())
    MRT_HIP(L.sample.alloc((size_t)std::max<uint32_t>(capacity, 1u) * (size_t)std::max(1, alloc_batch)));
    MRT_HIP(hipMemsetAsync(L.sample.p, 0, L.sample.bytes(), stream));
    if (shadow_planes && !need_thr) { if (int rc = alloc_planes(L)) return rc; }
    return MRT_OK;
}

// The G tile groups of this renderer's shard (renderer.h TileGroup) and their seed tables, built on the main stream (every lane forks from it at the next draw).
int Renderer::ensure_tile_groups(int G) {
    const int Bq = std::max(1, alloc_batch);
    if (tgroups_for == G && tgroups_batch == Bq) return MRT_OK;
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8, tiles = tiles_x * tiles_y;
    for (int g = 0; g < G; g++) {
        TileGroup &T = tgroups[g];
        T.rank = g * shard_world + shard_rank; T.world = G * shard_world;
        T.tiles_local = std::max(0, (tiles - T.rank + T.world - 1) / T.world);
        T.capacity = (uint32_t)T.tiles_local * 64u;
        T.owned = 0;
        for (int lt = 0; lt < T.tiles_local; lt++) {
            const int tile = lt * T.world + T.rank, ty = tile / tiles_x, tx = tile % tiles_x;
            T.owned += (uint64_t)std::min(8, width - tx * 8) * (uint64_t)std::min(8, height - ty * 8);
        }
        MRT_HIP(seeds_g[g].alloc((size_t)std::max<uint32_t>(T.capacity, 1u) * Bq));
        FrameParams sp{}; sp.width = width; sp.height = height; sp.shard_rank = T.rank; sp.shard_world = T.world; sp.tiles_x = tiles_x; sp.tiles_local = T.tiles_local; sp.capacity = T.capacity;
        if (T.capacity) hipLaunchKernelGGL(k_seed_slots, dim3(cdiv(T.capacity, 256), Bq), dim3(256), 0, stream, seeds_g[g].p, sp, seed);
        T.seeds = seeds_g[g].p;
    }
    tgroups_for = G; tgroups_batch = Bq;
    return MRT_OK;
}

int Renderer::set_shard(int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) { set_error("invalid shard"); return MRT_ERR_INVALID_ARGUMENT; }
    shard_rank = rank; shard_world = world;
    MRT_HIP(hipStreamSynchronize(stream));
    MRT_HIP(hipMemsetAsync(accum[0].p, 0, accum[0].bytes(), stream));
    MRT_HIP(hipMemsetAsync(accum[1].p, 0, accum[1].bytes(), stream));
    frame_index = 0; cur = 0;
    return alloc_queues();
}

int Renderer::render(int n_frames) {                                   // Renderer.draw(in:) :284-351, n times
    if (!scene) { set_error("renderer has no scene"); return MRT_ERR_STATE; }
    SceneView sv = scene->view();
    if (sv.light_count < 1) { set_error("scene has no lights (lightCount must be >= 1, Raytracing.metal:273)"); return MRT_ERR_STATE; }
    FrameParams fp{};
    fp.width = width; fp.height = height; fp.lightCount = light_count_limit > 0 ? std::min(light_count_limit, sv.light_count) : sv.light_count;
    fp.cam_pos = make_float4(camera.position.x, camera.position.y, camera.position.z, 0);
    fp.cam_right = make_float4(camera.right.x, camera.right.y, camera.right.z, 0);
    fp.cam_up = make_float4(camera.up.x, camera.up.y, camera.up.z, 0);
    fp.cam_fwd = make_float4(camera.forward.x, camera.forward.y, camera.forward.z, 0);
    fp.shard_rank = shard_rank; fp.shard_world = shard_world;
    fp.tiles_x = (width + 7) / 8; fp.tiles_local = tiles_local; fp.max_bounces = max_bounces;
    const uint32_t grid = std::max<uint32_t>(1u, (uint32_t)tiles_local);
    const bool two_level = sv.num_inst > 0;          // instanced scene: TLAS + BLASes walked by the same kernels (<TWO_LEVEL>)
    if (alloc_batch != batch_wanted()) {      // option (or the shard, under frame_batch = 0) changed since the buffers were sized
        MRT_HIP(hipStreamSynchronize(stream));
        const uint32_t keep_frame = frame_index; const int keep_cur = cur; const uint64_t keep_rendered = frames_rendered;
        DevBuf<float4> keep; MRT_HIP(keep.alloc(accum[cur].n));
        MRT_HIP(hipMemcpyAsync(keep.p, accum[cur].p, accum[cur].bytes(), hipMemcpyDeviceToDevice, stream));
        const MRTCamera keep_cam = camera;
        unsigned long long keep_totals[3] = {0, 0, 0};
        MRT_HIP(hipMemcpy(keep_totals, totals.p, sizeof keep_totals, hipMemcpyDeviceToHost));
        int rc = resize(width, height); if (rc) return rc;
        MRT_HIP(hipMemcpyAsync(totals.p, keep_totals, sizeof keep_totals, hipMemcpyHostToDevice, stream));
        MRT_HIP(hipMemcpyAsync(accum[keep_cur].p, keep.p, keep.bytes(), hipMemcpyDeviceToDevice, stream));
        MRT_HIP(hipStreamSynchronize(stream));
        frame_index = keep_frame; cur = keep_cur; frames_rendered = keep_rendered; frames_completed_known = keep_rendered; camera = keep_cam;
        return render(n_frames);
    }
    // the first draw sizes the lanes in use.  A lane's queues take ~163 B x pixels x frame_batch (1080p, 8-frame passes: 2.7 GB); when the
    // device cannot hold all the lanes asked for, the renderer runs on the ones it got (>= 1) instead of failing in the middle of a draw
    int F = std::max(1, std::min(frames_in_flight, MAX_FRAMES_IN_FLIGHT));
    // tile groups: a pass as G groups of tiles on G lanes (renderer.h TileGroup); the one-launch-per-frame mode has no passes to split.
    int G = 1;
    {
        const int cap0 = alloc_batch > DEFAULT_FRAME_BATCH ? std::min(alloc_batch, std::max(DEFAULT_FRAME_BATCH, (n_frames + 2) / 3)) : alloc_batch;
        const int np0 = (n_frames + std::max(1, cap0) - 1) / std::max(1, cap0), in_flight = std::max(1, std::min(F, np0));
        if (!megakernel && tile_groups != 1 && tiles_local >= 64) {
            const int lanes_target = std::max(F, 6);
            // by the draw (0): only passes of ONE frame — the launch-bound regimes: a frame alone 1.53 -> 1.42 ms as two or three groups (four: 1.53), three one-frame passes in flight
            // 0.85-0.91 -> 0.82-0.83 ms per frame as two groups each; a pass of seven or eight frames is better left whole (the driver's 20 frames as 3 x 2 groups: -4 %; profiles/r05_tile_groups.txt)
            G = tile_groups >= 2 ? tile_groups : cap0 > 1 ? 1 : in_flight == 1 ? 3 : in_flight <= 3 ? 2 : 1;
            (void)lanes_target;
            G = std::max(1, std::min(G, std::min(MAX_FRAMES_IN_FLIGHT / in_flight, tiles_local / 32)));
        }
        if (G > 1) F = in_flight * G;          // lanes this draw runs on: G per pass in flight
    }
    for (; lanes_ready < F; lanes_ready++) {
        const size_t need = lane_bytes() + (lanes_ready == 0 && !halton_tab.p ? (size_t)HALTON_TAB_DIMS * HALTON_TAB_SPAN * sizeof(float) : 0);      // (+ the renderer's one Halton table, allocated at its first bundled draw)
        size_t free_b = 0, total_b = 0;
        const bool fits = hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b > need + (size_t(1) << 30);      // keep 1 GiB of slack for the caller
        int rc = fits ? alloc_lane(lanes[lanes_ready]) : MRT_ERR_OUT_OF_MEMORY;
        if (rc == MRT_ERR_OUT_OF_MEMORY && lanes_ready >= 1) { (void)hipGetLastError(); release_lane(lanes[lanes_ready]); break; }
        if (rc == MRT_ERR_OUT_OF_MEMORY) { set_error("not enough device memory for one pass in flight: " + std::to_string(need >> 20) + " MiB of ray queues needed (" + std::to_string(width) + "x" + std::to_string(height) + ", frame_batch " + std::to_string(alloc_batch) + "); lower frame_batch"); return rc; }
        if (rc) return rc;
    }
    if (G > 1 && lanes_ready < F) { G = 1; F = std::max(1, std::min(std::min(frames_in_flight, MAX_FRAMES_IN_FLIGHT), lanes_ready)); }      // not enough memory for the groups' lanes: the plain form
    F = std::min(F, lanes_ready); lanes_used = F; groups_used = G;
    if (G > 1) { if (int rc = ensure_tile_groups(G)) return rc; }
    const int Fp = G > 1 ? F / G : F;          // passes in flight
    ext_used = 0;
    if (halton_table && frame_bundle && !megakernel && n_frames > 1) { if (int rc = ensure_halton_table(frame_index + sample_offset, (uint32_t)n_frames)) return rc; }      // (ahead of the fork: see there)
    MRT_HIP(hipEventRecord(ev_begin, stream));
    // fork: every lane starts after whatever the caller queued on the main stream (resize, camera, ...)
    MRT_HIP(hipEventRecord(ev_fork, stream));
    for (int k = 0; k < F; k++) MRT_HIP(hipStreamWaitEvent(lanes[k].stream, ev_fork, 0));
#ifdef MRT_DIAGNOSTICS
    // measuring aid of the diagnostics build only (tools/build_variant.sh diag "-DMRT_DIAGNOSTICS"; tools/archive/gpu_stage_ablation.sh): MRT_ABLATE=1 skips the primary launches, =2 the
    // bounce / shadow traversal launches — the other kernels then run on the stale but well-formed queues of an earlier pass, so their load is realistic and the frame time shows
    // what the skipped stage costs under overlap.  Images are garbage; the release library does not read the variable.
    static const int ablate = getenv("MRT_ABLATE") ? atoi(getenv("MRT_ABLATE")) : 0;
#else
    constexpr int ablate = 0;
#endif
    if (megakernel && (two_level || materials || sv.num_wnodes == 0)) {          // the one-launch-per-frame mode exists for flattened scenes on the 8-wide layout, diffuse kernel: say so instead of quietly rendering through the pipeline
        set_error(std::string("megakernel = 1 renders flattened scenes with the 8-wide layout and the reference's diffuse kernel only; this renderer has ") + (two_level ? "a two-level scene (scene option instancing = 1)" : materials ? "materials = 1" : "a scene without the 8-wide layout") + ": set megakernel = 0");
        return MRT_ERR_UNSUPPORTED;
    }
    const bool mega = megakernel;
    // passes larger than the default (sharded renderers ask for up to 32 frames so that a shard's launches stay large) never take more than a third of the draw:
    // a short draw keeps about three passes to run side by side (a rank of eight over 20 frames: 7.1 Grays/s as 7 + 7 + 6, 6.0 as one pass of 20)
    const int batch_cap = alloc_batch > DEFAULT_FRAME_BATCH ? std::min(alloc_batch, std::max(DEFAULT_FRAME_BATCH, (n_frames + 2) / 3)) : alloc_batch;
    const int batch_max = mega ? 1 : batch_cap;
    fp.npix = (uint32_t)((size_t)width * height); fp.capacity = capacity;
    hipEvent_t last_acc = nullptr;
    int pass = 0;
    const int n_passes = (n_frames + batch_max - 1) / batch_max;
    const int tail_from = (tail_accumulate && G == 1) ? n_passes - std::min(F, n_passes) : n_passes;      // passes from here on (each on a lane of its own) are accumulated together after the join (tile groups: every group accumulates its own pixels as it ends — nothing to serialise)
    hipEvent_t last_acc_g[MAX_TILE_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    const TileGroup self_group{shard_rank, shard_world, tiles_local, capacity, owned_pixels, seeds.p};
    AccGroup tail{}; tail.n = 0;
    for (int f = 0; f < n_frames; pass++) {
        // the draw's frames in passes of equal size (20 frames at frame_batch 8: 7 + 7 + 6, not 8 + 8 + 4 — the passes of a short draw run side by side and end together)
        int B = equal_passes ? std::min(batch_max, (n_frames - f + (n_passes - pass) - 1) / std::max(1, n_passes - pass)) : std::min(batch_max, n_frames - f);
        f += B;
        fp.batch = B;
        for (int g = 0; g < G; g++) {
        const TileGroup &TG = G > 1 ? tgroups[g] : self_group;
        if (G > 1 && TG.capacity == 0) continue;
        FrameLane &L = lanes[G > 1 ? (pass % Fp) * G + g : pass % F];
        hipStream_t st = L.stream;
        unsigned long long *bc = L.bounce_counts.p;                     // [bounce] {next rays (lo), shadow rays (hi)}, zero at frame start
        fp.frameIndex = frame_index;                                    // updateUniforms :216-229 (first frame of the batch)
        fp.sampleIndex = frame_index + sample_offset;
        // this group's tiles (G = 1: the renderer's own shard)
        fp.shard_rank = TG.rank; fp.shard_world = TG.world; fp.tiles_local = TG.tiles_local; fp.capacity = TG.capacity;
        const uint32_t capacity = TG.capacity, grid = std::max<uint32_t>(1u, (uint32_t)TG.tiles_local), grid_shade = std::max<uint32_t>(1u, cdiv(TG.capacity, SHADE_THREADS));
        const uint64_t owned_pixels = TG.owned;
        const uint32_t *const seeds_p = TG.seeds;
        bool used_planes = false;
        if (mega) {
            // one launch per frame on the pass's stream; frames are sequential (a path's last act is the running average with the previous target)
            const size_t stack_bytes = (size_t)scene->wide_depth * WIDE_STACK_LEVEL_BYTES;
            if (mega_slots_for_stack != stack_bytes) {
                int per_cu = 0, dev = 0; hipDeviceProp_t prop;
                MRT_HIP(hipGetDevice(&dev)); MRT_HIP(hipGetDeviceProperties(&prop, dev));
                MRT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_megakernel, 64, stack_bytes));
                mega_slots = std::max(1, per_cu) * prop.multiProcessorCount; mega_slots_for_stack = stack_bytes;
            }
            uint32_t *work = reinterpret_cast<uint32_t *>(bc + 64);      // its own word: the pipeline's per-bounce work counters must stay zero between passes
            for (int sub = 0; sub < B; sub++) {
                fp.frameIndex = frame_index + (uint32_t)sub; fp.sampleIndex = frame_index + (uint32_t)sub + sample_offset; fp.batch = 1;
                if (last_acc) { MRT_HIP(hipStreamWaitEvent(st, last_acc, 0)); last_acc = nullptr; }
                MRT_HIP(hipMemsetAsync(work, 0, 4, st));
                EvPair *ev = nullptr;
                if (ext_used < (int)ev_ext.size()) { ev_ext[ext_used].kind = MRT_KERNEL_TRACE; ev = &ev_ext[ext_used++]; }
                const uint32_t waves = (uint32_t)std::min<size_t>(std::max<size_t>(1, cdiv(capacity, 64)), (size_t)mega_slots);
                launch_timed(ev, k_megakernel, dim3(waves), dim3(64), stack_bytes, st, sv, fp, seeds_p, accum[cur].p, accum[1 - cur].p, work, totals.p, (uint32_t)owned_pixels);
                cur = 1 - cur;
            }
            MRT_HIP(hipEventRecord(L.accumulated, st));
            last_acc = L.accumulated;
            frame_index += (uint32_t)B; frames_rendered += (uint64_t)B;
            if (int rc = note_pass(st)) return rc;
            continue;
        }
        {
            // the pipeline: primary trace -> per bounce { shade, trace } ; bounce rays and shadow rays of a shade share one traversal launch
            const uint32_t grid_mixed = 2 * grid * (uint32_t)B;
            const bool on_wide = wide_bounce && sv.num_wnodes > 0;          // no 8-wide layout (scene option wide = 0, a tree deeper than WIDE_STACK_MAX): the rope kernels
            // two-level scenes walk TLAS and BLASes with the same kernels (traverse_wide_stream<true>); their LDS also parks the lanes' world rays
            const size_t stack_bytes = (size_t)scene->wide_depth * WIDE_STACK_LEVEL_BYTES + (two_level ? WIDE_WORLD_RAY_BYTES : 0);
            if (on_wide && persistent != 0 && slots_for_stack != stack_bytes) {       // wave slots of the chip for this kernel at this LDS size
                int per_cu = 0, dev = 0; hipDeviceProp_t prop;
                MRT_HIP(hipGetDevice(&dev)); MRT_HIP(hipGetDeviceProperties(&prop, dev));
                if (two_level) MRT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_mixed_wide_persist<true>, 64, stack_bytes));
                else MRT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_mixed_wide_persist<false>, 64, stack_bytes));
                if (!wave_slots_user) wave_slots = std::max(1, per_cu) * prop.multiProcessorCount;
                slots_for_stack = stack_bytes;
            }
            // traversal launches carry their own start/stop events (hipExtLaunchKernelGGL: the dispatch packet's timestamps, the
            // same clock rocprofv3 reads): plain hipEventRecord pairs on a stream also count the time a launch waits behind the
            // other frames in flight (+12 % at 12 frames)
            auto timed = [&](int kind) -> EvPair * { if (ext_used >= (int)ev_ext.size()) return nullptr; ev_ext[ext_used].kind = kind; return &ev_ext[ext_used++]; };
            fp.bounce = 0;
            fp.chain = (throughput_chain && !materials && max_bounces <= 3 && (uint64_t)scene->stats.instances * (uint64_t)std::max(1, scene->stats.max_submeshes) <= 65536ull) ? 1 : 0;
            if (!fp.chain && !L.thr[0].p) {           // this draw needs the throughput queues after all (materials, more than three bounces, a very large resource table)
                const size_t qcap = (size_t)this->capacity * (size_t)std::max(1, alloc_batch);          // (a lane's buffers are sized for the renderer's whole shard, whatever tile group uses it now)
                for (int k = 0; k < 2; k++) MRT_HIP(L.thr[k].alloc(qcap));
            }
            const uint32_t rpw_p = stream_rays_per_wave((size_t)capacity * B), rpw_m = stream_rays_per_wave(2 * (size_t)capacity * B);
            float4 *const dirs = L.rayB[1].p;
            // shadow planes: contribution per pixel and bounce + one byte per shadow ray that got through, instead of the contribution queue and the read-modify-write of the sample buffer
            const bool planes_pass = shadow_planes != 0 && fp.chain && on_wide && !materials && max_bounces <= PLANES_MAX_BOUNCES && !ablate;
            if (planes_pass) {
                if (!L.f_lit.p) { if (int rc = alloc_planes(L)) return rc; }
                MRT_HIP(hipMemsetAsync(L.f_lit.p, 0, 4 * (size_t)capacity * (size_t)B, st));
            }
            used_planes = planes_pass;
            // two-level scenes: the binned walk (TLAS pass + BLAS pass over (ray, instance) pairs) for the bounce / shadow rays of planes passes
            const bool pairs_pass = two_level && planes_pass && tl_pairs != 0 && sv.tri_packet != nullptr;
            const size_t pair_cap = 2 * (size_t)this->capacity * (size_t)std::max(1, alloc_batch);          // one pair per virtual ray of the combined queue; a push beyond it walks its instance in place
            if (pairs_pass && !L.pairs.p) MRT_HIP(L.pairs.alloc(std::max<size_t>(PairQueue::WORDS * pair_cap, 1)));
            // few instances: the TLAS pass has no tree (every lane visits every instance: nothing diverges; two_level_passes.h k_tl_top_flat)
            const bool tl_flat = pairs_pass && sv.num_inst <= TL_FLAT_MAX_INSTANCES && tl_pairs != 2;
            const uint32_t pair_cap_used = (uint32_t)std::min<size_t>(tl_pair_cap > 0 ? std::min<size_t>((size_t)tl_pair_cap, pair_cap) : pair_cap, 0xFFFFFFFFu);
            // the primary trace inside shade(0): flattened scenes, planes passes
            // (not for one frame alone on the chip, fuse_primary = 1: there the primary kernel's 48 registers and 64-thread workgroups fill the chip better than shade's 76 and 256 — 1.71 against 1.81 ms;
            // fuse_primary = 2 fuses always)
            // which layout the primary rays of a flattened scene walk: the 8-wide one when the scene has it (primary_wide = 2, default: one ray per lane inside shade(0) or in its own launch;
            // = 1: the stream kernel with lane refill, A/B), the rope layout otherwise — or on request (primary_wide = 0; needs scene option rope = 1)
            const bool prim_rope = !two_level && (!sv.num_wnodes || primary_wide == 0);
            if (prim_rope && sv.num_nodes == 0 && sv.num_tris != 0) { set_error("primary_wide = 0 walks the rope layout: commit the scene with scene option rope = 1"); return MRT_ERR_STATE; }
            if (!on_wide && !two_level && sv.num_nodes == 0 && sv.num_tris != 0) { set_error("wide_bounce = 0 walks the rope layout: commit the scene with scene option rope = 1"); return MRT_ERR_STATE; }
            const bool trace0_pass = planes_pass && fuse_primary != 0 && primary_wide != 1 && !(two_level && primary_wide == 0) && (fuse_primary == 2 || F > 1 || B > 1 || two_level);      // (two-level scenes always: their own-launch form is the stream kernel, 0.88 ms for one 1080p frame of dragon x 4)
            const bool trace0_wide = trace0_pass && !prim_rope;          // (planes_pass implies the 8-wide layout)
            const bool trace0_hint = primary_hint && (!two_level || (sv.num_inst <= 255u && scene->wpackets.n / WPK < ((size_t)1 << 24)));      // two-level: the hint is (packet | instance << 24)
            fp.wide_stack_words = (uint32_t)((size_t)scene->wide_depth * WIDE_STACK_LEVEL_BYTES / 4);
            if (!planes_pass && !L.scon.p) MRT_HIP(L.scon.alloc((size_t)this->capacity * (size_t)std::max(1, alloc_batch)));
            if ((ablate & 1) || trace0_pass) {}
            else if (two_level && on_wide) {
                // the hint of two-level scenes is (packet | instance << 24): scenes of at most 255 instances and 2^24 packets
                const bool seeded = primary_hint && sv.num_inst <= 255u && scene->wpackets.n / WPK < ((size_t)1 << 24);
                if (seeded) launch_timed(timed(MRT_KERNEL_PRIMARY), k_trace_primary_wide_stream<true, true>, dim3(cdiv(capacity, rpw_p), B), dim3(64), stack_bytes, st, sv, fp, seeds_p, L.hits.p, dirs, capacity, rpw_p, hint.p);
                else launch_timed(timed(MRT_KERNEL_PRIMARY), k_trace_primary_wide_stream<true, false>, dim3(cdiv(capacity, rpw_p), B), dim3(64), stack_bytes, st, sv, fp, seeds_p, L.hits.p, dirs, capacity, rpw_p, (uint32_t *)nullptr);
            }
            else if (primary_wide == 1 && sv.num_wnodes && !two_level) {
                if (primary_hint) launch_timed(timed(MRT_KERNEL_PRIMARY), k_trace_primary_wide_stream<false, true>, dim3(cdiv(capacity, rpw_p), B), dim3(64), stack_bytes, st, sv, fp, seeds_p, L.hits.p, dirs, capacity, rpw_p, hint.p);
                else launch_timed(timed(MRT_KERNEL_PRIMARY), k_trace_primary_wide_stream<false, false>, dim3(cdiv(capacity, rpw_p), B), dim3(64), stack_bytes, st, sv, fp, seeds_p, L.hits.p, dirs, capacity, rpw_p, (uint32_t *)nullptr);
            }
            else if (two_level) launch_timed(timed(MRT_KERNEL_PRIMARY), k_trace_primary<true>, dim3(grid, B), dim3(64), 0, st, sv, fp, seeds_p, L.hits.p, dirs, (uint32_t *)nullptr);
            else if (!prim_rope) launch_timed(timed(MRT_KERNEL_PRIMARY), k_trace_primary<false, true>, dim3(grid, B), dim3(64), (size_t)scene->wide_depth * WIDE_STACK_LEVEL_BYTES, st, sv, fp, seeds_p, L.hits.p, dirs, primary_hint ? hint.p : (uint32_t *)nullptr);
            else launch_timed(timed(MRT_KERNEL_PRIMARY), k_trace_primary<false>, dim3(grid, B), dim3(64), 0, st, sv, fp, seeds_p, L.hits.p, dirs, primary_hint ? hint.p : (uint32_t *)nullptr);
            int q = 0;                                                  // shade(b) writes next rays into queue q
            for (int b = 0; b < max_bounces; b++) {
                fp.bounce = b;
                const unsigned long long *cin = b == 0 ? nullptr : bc + (b - 1);
                // bounce 0 reads no ray queue (it regenerates the primary ray); bounce b > 0 reads the queue shade(b-1) wrote
                // bounce 0: one grid row per sub-frame of the batch over the primary slots; later bounces: the compact queue of the whole batch
                const bool pack = shade_pack && b > 0;          // bounces >= 1 read a queue half of whose rays missed: its hits are compacted in LDS and shaded on full waves (k_shade_pack)
                // entries per packing workgroup: SHADE_PACK_RANGE when the queue is long, less when that would leave fewer than ~2048 workgroups (a one-frame pass, a tile group, a shard) — never less than two rounds' worth
                fp.pack_range = (uint32_t)std::min<size_t>(SHADE_PACK_RANGE, std::max<size_t>(2 * SHADE_THREADS, (size_t)capacity * B / 2048 / SHADE_THREADS * SHADE_THREADS));
                fp.frame_bundle = (frame_bundle && b == 0 && trace0_wide && B > 1) ? 1 : 0;
                if (fp.frame_bundle) {
                    fp.bundle_groups = ((uint32_t)B + 7u) / 8u; fp.bundle_w = ((uint32_t)B + fp.bundle_groups - 1u) / fp.bundle_groups;
                    fp.bundle_per_wave = 64u / fp.bundle_w; fp.bundle_magic = (65536u + fp.bundle_w - 1u) / fp.bundle_w;
                }
                fp.halton_tab = nullptr; fp.halton_w0 = 0; fp.halton_n = 0;
                if (b == 0 && fp.frame_bundle && fp.bundle_w >= 4u && halton_table && halton_tab.p) {          // a wave reads bundle_w consecutive values per load: the table pays from four on (indices outside its window: the recurrence)
                    fp.halton_tab = halton_tab.p; fp.halton_w0 = halton_w0; fp.halton_n = HALTON_TAB_SPAN;
                }
                const dim3 gs = b == 0 ? (fp.frame_bundle ? dim3(cdiv(cdiv((size_t)capacity * fp.bundle_groups, fp.bundle_per_wave) * 64, SHADE_THREADS), 1) : dim3(grid_shade, B)) : dim3(cdiv((size_t)capacity * B, pack ? fp.pack_range : (uint32_t)SHADE_THREADS));
                using ShadeKernel = void (*)(SceneView, FrameParams, const uint32_t *, const float4 *, const float4 *, const float4 *, const float4 *, const unsigned long long *, uint32_t, float4 *, float4 *, float4 *, float4 *, float4 *, float4 *,
                                             unsigned long long *, float4 *, float4 *, uint32_t *);
                const bool trace0_tl = trace0_wide && b == 0 && two_level;
                const ShadeKernel shade_kernel = pack ? (materials ? (ShadeKernel)k_shade_pack<true, false, false, false>
                                                          : pairs_pass ? (ShadeKernel)k_shade_pack<false, true, true, true>
                                                          : planes_pass ? (ShadeKernel)k_shade_pack<false, true, true, false>
                                                          : fp.chain ? (ShadeKernel)k_shade_pack<false, true, false, false> : (ShadeKernel)k_shade_pack<false, false, false, false>)
                                                  : materials ? (ShadeKernel)k_shade<true, false, false, false>
                                                  : (pairs_pass && b > 0) ? (ShadeKernel)k_shade<false, true, true, true>
                                                  : trace0_tl ? (ShadeKernel)k_shade_primary<3>
                                                  : (trace0_wide && b == 0) ? (ShadeKernel)k_shade_primary<2>
                                                  : (trace0_pass && b == 0) ? (ShadeKernel)k_shade_primary<1>
                                                  : planes_pass ? (ShadeKernel)k_shade<false, true, true, false>
                                                  : fp.chain ? (ShadeKernel)k_shade<false, true, false, false> : (ShadeKernel)k_shade<false, false, false, false>;
                float4 *const con_b = !planes_pass ? L.scon.p : b == 0 ? L.sample.p : L.f_con[b - 1].p;         // PLANES: this bounce's contribution plane in place of the queue
                uint8_t *const lit_b = planes_pass ? L.f_lit.p + b : nullptr;
                ShadeIO io{};
                io.seeds = seeds_p; io.rayA = L.rayA[1 - q].p; io.rayB = L.rayB[1 - q].p; io.thr = L.thr[1 - q].p; io.hits = L.hits.p;          // (two-level, binned, b > 0: the 64-bit keys of the TLAS / BLAS passes, in the hit buffer)
                io.count_in = cin; io.capacity = capacity;
                io.nrayA = L.rayA[q].p; io.nrayB = L.rayB[q].p; io.nthr = L.thr[q].p; io.srayA = L.srayA.p; io.srayB = L.srayB.p; io.scon = con_b; io.count_out = bc + b;
                io.sample_primary = b == 0 ? L.sample.p : nullptr; io.sample = L.sample.p; io.hint = (b == 0 && trace0_pass && trace0_hint) ? hint.p : nullptr;
                uint32_t *const pc = reinterpret_cast<uint32_t *>(bc + 65 + b);          // two-level, binned: {pairs queued, work counter of the BLAS pass}
                const size_t shade_lds = (trace0_wide && b == 0) ? (size_t)SHADE_WAVES * (scene->wide_depth * WIDE_STACK_LEVEL_BYTES + (MRT_LANE_HIT_LDS ? 1024 : 0)) : 0;
                launch_timed(timed(MRT_KERNEL_SHADE), shade_kernel, gs, dim3(SHADE_THREADS), shade_lds, st, sv, fp, io.seeds, io.rayA, io.rayB, io.thr, io.hits, io.count_in, io.capacity, io.nrayA, io.nrayB, io.nthr, io.srayA, io.srayB, io.scon,
                             io.count_out, io.sample_primary, io.sample, io.hint);
                // persistent = 2 (auto): pull chunks when every wave slot would otherwise own >= 1024 rays (4-frame passes at 1080p: +7...+11 % with
                // one stream, +2.5 % with 12); one-frame launches keep the static split (384 rays per wave, no atomics: 3 frames in flight 6.5 vs 5.4 Grays/s)
                // [r3] smaller launches pull as well when five or more passes are in flight (6 lanes x one-frame passes: 9.33 against 8.76 Grays/s; a rank of eight over 240 frames
                // in 8-frame passes: 9.43 against 8.56); with one to three passes in flight they do better on the even static split (one frame alone 1.51 against 1.76 ms, 3 x 1 frame
                // 7.63 against 7.20 Grays/s, a rank of eight over the driver's 20 frames 6.65 against 5.73): stream_even below
                const bool pull = persistent == 1 || (persistent == 2 && (2 * (size_t)capacity * B >= (size_t)wave_slots * 1024 || (std::min(F, G > 1 ? Fp * G : n_passes) >= 5 && 2 * (size_t)capacity * B >= (size_t)wave_slots * 256)));      // (below 256 slots per wave slot — Cornell 256^2 in 8-frame passes — the even split: 6.46 against 5.40 Grays/s)
                if (ablate & 2) {}
                else if (pairs_pass) {
                    const size_t slots = 2 * (size_t)capacity * B;
                    const uint32_t chunk = (uint32_t)std::min<size_t>((size_t)persist_chunk, std::max<size_t>(64, slots / ((size_t)wave_slots * 4) / 64 * 64));
                    const size_t grid_slots = (!wave_slots_user && (n_frames + batch_max - 1) / batch_max >= 2 * F) ? (size_t)std::max(1, wave_slots / 2) : (size_t)wave_slots;
                    const uint32_t waves = (uint32_t)std::max<size_t>(1, std::min<size_t>(cdiv(slots, chunk), grid_slots));
                    unsigned long long *const keys = reinterpret_cast<unsigned long long *>(L.hits.p);
                    // few instances: the TLAS pass without a tree; many: the stream walk of the 8-wide TLAS
                    if (tl_flat)
                        launch_timed(timed(MRT_KERNEL_TRACE), k_tl_top_flat, dim3((uint32_t)std::max<size_t>(1, std::min<size_t>(cdiv(slots, 64), 2 * grid_slots))), dim3(64), stack_bytes + 8, st, sv, L.rayA[q].p, L.rayB[q].p, keys, L.srayA.p, L.srayB.p,
                                     (const unsigned long long *)(bc + b), lit_b, L.pairs.p, pc, pair_cap_used, (uint32_t)(stack_bytes / 4));
                    else
                    launch_timed(timed(MRT_KERNEL_TRACE), k_tl_top, dim3(waves), dim3(64), stack_bytes + 8, st, sv, L.rayA[q].p, L.rayB[q].p, keys, L.srayA.p, L.srayB.p,
                                 (const unsigned long long *)(bc + b), reinterpret_cast<uint32_t *>(bc + 32 + b), chunk, lit_b, L.pairs.p, pc, pair_cap_used, (uint32_t)(stack_bytes / 4));
                    // the pairs' count is on the device: the launch has the wave slots it may use and the surplus leaves at once
                    launch_timed(timed(MRT_KERNEL_TRACE), k_tl_blas, dim3((uint32_t)grid_slots), dim3(64), (size_t)scene->wide_depth * WIDE_STACK_LEVEL_BYTES, st, sv, L.rayA[q].p, L.rayB[q].p, keys, L.srayA.p, L.srayB.p,
                                 (const unsigned long long *)(bc + b), pc + 1, 256u, lit_b, (const uint4 *)L.pairs.p, (const uint32_t *)pc, pair_cap_used);
                }
                else if (on_wide && pull) {
                    // rays per pull: at least four pulls per wave slot on a queue of this size (so that the launch ends evenly), at most
                    // persist_chunk; 128-ray pulls of a one-frame launch are ~78 atomics per microsecond on the one counter word (limit ~88)
                    const size_t slots = 2 * (size_t)capacity * B;
                    const uint32_t chunk = (uint32_t)std::min<size_t>((size_t)persist_chunk, std::max<size_t>(128, slots / ((size_t)wave_slots * 4) / 64 * 64));
                    // a long call (every lane gets several passes) runs its traversal launches on HALF the wave slots: the other passes' shade, primary
                    // and accumulate blocks then find free slots instead of queueing behind persistent waves that only leave when their queue is empty
                    // (measured, 240 steps: 4 lanes 9.86 -> 10.06, 6 lanes 10.08 -> 10.32, 12 lanes 10.38 -> 10.55 Grays/s; a 20-step call, whose five
                    // passes move in lock step, loses 3 % and keeps the full grid)
                    const size_t grid_slots = (!wave_slots_user && (n_frames + batch_max - 1) / batch_max >= 2 * F) ? (size_t)std::max(1, wave_slots / 2) : (size_t)wave_slots;
                    const uint32_t waves = (uint32_t)std::min<size_t>(cdiv(slots, chunk), grid_slots);
#ifdef MRT_WAVE_TIMES
                    const uint32_t chunk_arg = chunk | ((uint32_t)b << 24);
#else
                    const uint32_t chunk_arg = chunk;
#endif
                    if (!two_level && planes_pass && hit_lds) {
                        // the variant with the hit words in LDS: its own LDS size, hence its own count of wave slots
                        const size_t lds_x = (size_t)HIT_LDS_WORDS * 4 + stack_bytes;
                        const int key = (int)(lds_x & 0xFFFFFF);
                        if (slots_x_key != key) {
                            int per_cu = 0, dev = 0; hipDeviceProp_t prop;
                            MRT_HIP(hipGetDevice(&dev)); MRT_HIP(hipGetDeviceProperties(&prop, dev));
                            MRT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_mixed_wide_persist_x, 64, lds_x));
                            if (per_cu < 1) { set_error("hit_lds: the traversal kernel does not fit a compute unit with " + std::to_string(lds_x) + " bytes of LDS"); return MRT_ERR_UNSUPPORTED; }
                            wave_slots_x = per_cu * prop.multiProcessorCount; slots_x_key = key;
                        }
                        const size_t ws = wave_slots_user ? (size_t)wave_slots : (size_t)wave_slots_x;
                        const uint32_t chunk_x = (uint32_t)std::min<size_t>((size_t)persist_chunk, std::max<size_t>(128, slots / (ws * 4) / 64 * 64));
                        const size_t grid_slots_x = (!wave_slots_user && (n_frames + batch_max - 1) / batch_max >= 2 * F) ? std::max<size_t>(1, ws / 2) : ws;
                        const uint32_t waves_x = (uint32_t)std::max<size_t>(1, std::min<size_t>(cdiv(slots, chunk_x), grid_slots_x));
                        launch_timed(timed(MRT_KERNEL_TRACE), k_trace_mixed_wide_persist_x, dim3(waves_x), dim3(64), lds_x, st, sv, (const float4 *)L.rayA[q].p, (const float4 *)L.rayB[q].p, L.hits.p, (const float4 *)L.srayA.p, (const float4 *)L.srayB.p,
                                     (const unsigned long long *)(bc + b), reinterpret_cast<uint32_t *>(bc + WORK_COUNTERS + (size_t)b * WORK_COUNTERS_PER_BOUNCE), chunk_x, lit_b, xcd_counters ? (uint32_t)B : 0u);
                    }
                    else if (two_level) launch_timed(timed(MRT_KERNEL_TRACE), k_trace_mixed_wide_persist<true>, dim3(std::max(1u, waves)), dim3(64), stack_bytes, st, sv, L.rayA[q].p, L.rayB[q].p, L.hits.p, L.srayA.p, L.srayB.p, L.scon.p,
                                 (const unsigned long long *)(bc + b), L.sample.p, reinterpret_cast<uint32_t *>(bc + WORK_COUNTERS + (size_t)b * WORK_COUNTERS_PER_BOUNCE), chunk_arg, lit_b, xcd_counters ? (uint32_t)B : 0u);
                    else launch_timed(timed(MRT_KERNEL_TRACE), k_trace_mixed_wide_persist<false>, dim3(std::max(1u, waves)), dim3(64), stack_bytes, st, sv, L.rayA[q].p, L.rayB[q].p, L.hits.p, L.srayA.p, L.srayB.p, L.scon.p,
                                 (const unsigned long long *)(bc + b), L.sample.p, reinterpret_cast<uint32_t *>(bc + WORK_COUNTERS + (size_t)b * WORK_COUNTERS_PER_BOUNCE), chunk_arg, lit_b, xcd_counters ? (uint32_t)B : 0u);
                }
                else if (on_wide) {
                    const size_t slots_m = 2 * (size_t)capacity * B;
                    // a shard's launches (a rank of eight over the driver's 20 frames: three passes of its 1/8 of the tiles in flight) do better with ONE round of waves that take the queue's
                    // 64-ray batches round-robin — every rank of eight timed: 2.00 against 2.21 ms on average, the slowest 2.13-2.18 against 2.46-2.50 (profiles/r05_shard_stride.txt); a whole
                    // image's one-frame launches do not (one frame alone 1.38 = 1.38 ms, three in flight 0.775 against 0.765)
                    const bool takes_x = !two_level && planes_pass && hit_lds;          // k_trace_mixed_wide_stream_x: the one kernel that deals batches round-robin (BatchStride)
                    const bool shard_auto = stream_stride == 2 && stream_even == 200 && this->shard_world > 1 && G == 1 && takes_x;          // (one round of waves was measured with the strided deal only)
                    const bool strided = stream_stride == 1 || shard_auto;
                    const int even_pct = shard_auto ? 100 : stream_even;
                    const uint32_t even = even_pct > 0 ? (uint32_t)std::max<size_t>(1, std::min<size_t>(cdiv(slots_m, 64), (size_t)wave_slots * (size_t)even_pct / 100)) : 0u;     // stream_even: percent of the wave slots
                    const dim3 grid_s(even ? even : cdiv(slots_m, rpw_m));
#ifdef MRT_WAVE_TIMES
                    const uint32_t rpw_m_arg = rpw_m | ((uint32_t)b << 24);
#else
                    const uint32_t rpw_m_arg = rpw_m;
#endif
                    if (takes_x) launch_timed(timed(MRT_KERNEL_TRACE), k_trace_mixed_wide_stream_x, grid_s, dim3(64), stack_bytes + HIT_LDS_WORDS * 4, st, sv, (const float4 *)L.rayA[q].p, (const float4 *)L.rayB[q].p, L.hits.p, (const float4 *)L.srayA.p, (const float4 *)L.srayB.p, (const unsigned long long *)(bc + b), rpw_m_arg, lit_b, even | (strided ? 0x80000000u : 0u));
                    else if (two_level) launch_timed(timed(MRT_KERNEL_TRACE), k_trace_mixed_wide_stream<true>, grid_s, dim3(64), stack_bytes, st, sv, L.rayA[q].p, L.rayB[q].p, L.hits.p, L.srayA.p, L.srayB.p, L.scon.p, (const unsigned long long *)(bc + b), L.sample.p, rpw_m, lit_b, even);
                    else launch_timed(timed(MRT_KERNEL_TRACE), k_trace_mixed_wide_stream<false>, grid_s, dim3(64), stack_bytes, st, sv, L.rayA[q].p, L.rayB[q].p, L.hits.p, L.srayA.p, L.srayB.p, L.scon.p, (const unsigned long long *)(bc + b), L.sample.p, rpw_m, lit_b, even);
                }
                else if (two_level) launch_timed(timed(MRT_KERNEL_TRACE), k_trace_mixed<true>, dim3(grid_mixed), dim3(64), 0, st, sv, L.rayA[q].p, L.rayB[q].p, L.hits.p, L.srayA.p, L.srayB.p, L.scon.p, (const unsigned long long *)(bc + b), L.sample.p);
                else launch_timed(timed(MRT_KERNEL_TRACE), k_trace_mixed<false>, dim3(grid_mixed), dim3(64), 0, st, sv, L.rayA[q].p, L.rayB[q].p, L.hits.p, L.srayA.p, L.srayB.p, L.scon.p, (const unsigned long long *)(bc + b), L.sample.p);
                q = 1 - q;
            }
        }
        // accumulation is the only frame-to-frame dependency (prev target = the previous frame's output)
        const bool deferred = used_planes && pass >= tail_from && tail.n < MAX_FRAMES_IN_FLIGHT;
        if (deferred) {
            AccPass &P = tail.p[tail.n++];
            P.con0 = L.sample.p; P.con1 = L.f_con[0].p; P.con2 = L.f_con[1].p; P.lit = L.f_lit.p; P.counts = bc; P.frameIndex = fp.frameIndex; P.batch = B; P.primary = (uint32_t)(owned_pixels * (uint64_t)B); P.pad = 0;
            MRT_HIP(hipEventRecord(L.accumulated, st));       // (here: traced — the join below waits for it, the accumulation follows on the main stream)
            frame_index += (uint32_t)B; frames_rendered += (uint64_t)B;
            continue;
        }
        if (G > 1) { if (last_acc_g[g] && last_acc_g[g] != L.accumulated) MRT_HIP(hipStreamWaitEvent(st, last_acc_g[g], 0)); }      // this group's pixels of the previous target: written by the same group of the pass before
        else if (last_acc) MRT_HIP(hipStreamWaitEvent(st, last_acc, 0));
        {
            EvPair *ev = nullptr;
            if (ext_used < (int)ev_ext.size()) { ev_ext[ext_used].kind = MRT_KERNEL_ACCUMULATE; ev = &ev_ext[ext_used++]; }
            if (used_planes) launch_timed(ev, k_accumulate_planes, dim3(grid), dim3(64), 0, st, fp, (const float4 *)L.sample.p, (const float4 *)L.f_con[0].p, (const float4 *)L.f_con[1].p, (const uint8_t *)L.f_lit.p,
                                          (const float4 *)accum[cur].p, accum[1 - cur].p, bc, totals.p, (uint32_t)(owned_pixels * (uint64_t)B));
            else launch_timed(ev, k_accumulate, dim3(grid), dim3(64), 0, st, fp, L.sample.p, accum[cur].p, accum[1 - cur].p, bc, totals.p, (uint32_t)(owned_pixels * (uint64_t)B));
        }
        MRT_HIP(hipEventRecord(L.accumulated, st));
        if (G > 1) { last_acc_g[g] = L.accumulated; continue; }
        last_acc = L.accumulated;
        cur = 1 - cur;                                                  // ping-pong swap :332-334 (once per batch: the batch's frames are applied in one kernel)
        frame_index += (uint32_t)B; frames_rendered += (uint64_t)B;
        if (int rc = note_pass(st)) return rc;
        }       // tile groups of the pass
        if (G > 1) { cur = 1 - cur; frame_index += (uint32_t)B; frames_rendered += (uint64_t)B; }      // every group read accum[cur] and wrote accum[1 - cur] at its own pixels
    }
    // join: the main stream continues after every lane has drained
    for (int k = 0; k < (G > 1 ? std::min(Fp, pass) * G : std::min(F, pass)); k++) MRT_HIP(hipStreamWaitEvent(stream, lanes[k].accumulated, 0));
    if (G > 1) { if (int rc = note_pass(stream)) return rc; }      // (tile groups: completion is reported per draw)
    if (tail.n > 0) {
        EvPair *ev = nullptr;
        if (ext_used < (int)ev_ext.size()) { ev_ext[ext_used].kind = MRT_KERNEL_ACCUMULATE; ev = &ev_ext[ext_used++]; }
        launch_timed(ev, k_accumulate_planes_group, dim3(grid), dim3(64), 0, stream, fp, tail, (const float4 *)accum[cur].p, accum[1 - cur].p, totals.p);
        cur = 1 - cur;
        if (int rc = note_pass(stream)) return rc;
    }
    MRT_HIP(hipEventRecord(ev_end, stream));
    MRT_HIP(hipGetLastError());
    pending_timing = true;
    return MRT_OK;
}

#ifdef MRT_DEBUG_BOUNDS
static int check_bounds_record() {
    uint32_t v = 0;
    MRT_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bounds_violation), 4));
    if (v) {
        static const char *kinds[] = {"?", "8-wide node", "triangle packet", "wtlas_index slot", "instance id"};
        const uint32_t zero = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bounds_violation), &zero, 4);
        set_error(std::string("MRT_DEBUG_BOUNDS: the stream traversal followed an index outside its array: ") + kinds[std::min(v >> 28, 4u)] + " " + std::to_string(v & 0x0FFFFFFFu));
        return MRT_ERR_STATE;
    }
    return MRT_OK;
}
#else
static int check_bounds_record() { return MRT_OK; }
#endif

int Renderer::wait() {
    MRT_HIP(hipStreamSynchronize(stream));
    if (int rc = check_bounds_record()) return rc;
    { uint64_t done = 0; if (int rc = poll_completed(&done)) return rc; }
    if (pending_timing) {
        float ms = 0;
        MRT_HIP(hipEventElapsedTime(&ms, ev_begin, ev_end));
        ms_last = ms;
        float ext = 0; uint32_t next = 0;
        kernel_times = MRTKernelTimes{};
        for (int k = 0; k < ext_used; k++) {
            float e = 0; MRT_HIP(hipEventElapsedTime(&e, ev_ext[k].a, ev_ext[k].b));
            const int kind = ev_ext[k].kind;
            kernel_times.ms[kind] += e; kernel_times.launches[kind]++;
            if (kind == MRT_KERNEL_PRIMARY || kind == MRT_KERNEL_TRACE) { ext += e; next++; }
        }
        ms_extend_last = ext; extend_launches_last = next;
        pending_timing = false;
    }
    return MRT_OK;
}

int Renderer::read_accum(float *rgba, size_t nbytes) {
    if (nbytes != (size_t)width * height * 16) { set_error("read_accum: nbytes must be width*height*16"); return MRT_ERR_INVALID_ARGUMENT; }
    int rc = wait(); if (rc) return rc;
    MRT_HIP(hipMemcpy(rgba, accum[cur].p, nbytes, hipMemcpyDeviceToHost));
    return MRT_OK;
}
int Renderer::copy_accum_to_device(void *dptr, size_t nbytes) {
    if (nbytes != (size_t)width * height * 16) { set_error("copy_accum_to_device: nbytes must be width*height*16"); return MRT_ERR_INVALID_ARGUMENT; }
    MRT_HIP(hipMemcpyAsync(dptr, accum[cur].p, nbytes, hipMemcpyDeviceToDevice, stream));
    return MRT_OK;
}
int Renderer::write_accum_from_device(const void *dptr, size_t nbytes) {
    if (nbytes != (size_t)width * height * 16) { set_error("write_accum_from_device: nbytes must be width*height*16"); return MRT_ERR_INVALID_ARGUMENT; }
    MRT_HIP(hipMemcpyAsync(accum[cur].p, dptr, nbytes, hipMemcpyDeviceToDevice, stream));
    return MRT_OK;
}
uint32_t shard_tiles(int width, int height, int rank, int world) {
    const int tiles = ((width + 7) / 8) * ((height + 7) / 8);
    return (uint32_t)std::max(0, (tiles - rank + world - 1) / world);
}
int unpack_tiles_into(float4 *image, int width, int height, const void *compact, size_t nbytes, int rank, int world, hipStream_t st) {
    if (world < 1 || rank < 0 || rank >= world) { set_error("unpack_tiles: invalid shard"); return MRT_ERR_INVALID_ARGUMENT; }
    const uint32_t tl = shard_tiles(width, height, rank, world);
    if (nbytes != (size_t)tl * 64 * sizeof(float4)) { set_error("unpack_tiles: nbytes must be tiles x 64 x 16 for that shard (mrt_renderer_shard_tiles)"); return MRT_ERR_INVALID_ARGUMENT; }
    if (tl) hipLaunchKernelGGL(k_tiles<false>, dim3(cdiv((size_t)tl * 64, 256)), dim3(256), 0, st, image, width, height, (width + 7) / 8, rank, world, tl, (float4 *)const_cast<void *>(compact));
    MRT_HIP(hipGetLastError());
    return MRT_OK;
}
int Renderer::pack_owned_tiles(void *dptr, size_t nbytes) {
    const uint32_t tl = shard_tiles(width, height, shard_rank, shard_world);
    if (nbytes != (size_t)tl * 64 * sizeof(float4)) { set_error("pack_owned_tiles: nbytes must be tiles x 64 x 16 for this renderer's shard (mrt_renderer_shard_tiles)"); return MRT_ERR_INVALID_ARGUMENT; }
    if (tl) hipLaunchKernelGGL(k_tiles<true>, dim3(cdiv((size_t)tl * 64, 256)), dim3(256), 0, stream, accum[cur].p, width, height, (width + 7) / 8, shard_rank, shard_world, tl, (float4 *)dptr);
    MRT_HIP(hipGetLastError());
    return MRT_OK;
}
int Renderer::unpack_tiles(const void *dptr, size_t nbytes, int rank, int world) { return unpack_tiles_into(accum[cur].p, width, height, dptr, nbytes, rank, world, stream); }
int Renderer::read_tonemapped(uint8_t *rgba, size_t nbytes) {
    if (nbytes != (size_t)width * height * 4) { set_error("read_tonemapped: nbytes must be width*height*4"); return MRT_ERR_INVALID_ARGUMENT; }
    DevBuf<uchar4> tmp; MRT_HIP(tmp.alloc((size_t)width * height));
    hipLaunchKernelGGL(k_tonemap, dim3(cdiv(width, 16), cdiv(height, 16)), dim3(16, 16), 0, stream, accum[cur].p, width, height, tmp.p);
    int rc = wait(); if (rc) return rc;
    MRT_HIP(hipMemcpy(rgba, tmp.p, nbytes, hipMemcpyDeviceToHost));
    return MRT_OK;
}

int Renderer::stats(MRTRenderStats *out) {
    int rc = wait(); if (rc) return rc;
    unsigned long long t[4];
    MRT_HIP(hipMemcpy(t, totals.p, sizeof t, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->frames = frames_rendered; out->closest_rays = t[0]; out->shadow_rays = t[1]; out->primary_rays = t[2];
    // 96 B per closest ray, 72 B per shadow ray, 36 B per pixel (20 on frame 0), + one read of the scene per frame
    uint64_t px = owned_pixels;
    uint64_t per_frame_px = px * 36;
    out->bytes_alg = t[0] * 96 + t[1] * 72 + frames_rendered * per_frame_px + frames_rendered * scene->stats.scene_bytes;
    out->ms_gpu_last = ms_last; out->ms_extend_last = ms_extend_last; out->extend_launches_last = extend_launches_last;
    return MRT_OK;
}
int Renderer::reset_stats() {
    MRT_HIP(hipMemsetAsync(totals.p, 0, totals.bytes(), stream));
    MRT_HIP(hipStreamSynchronize(stream));       // frames_completed counts from here: nothing of the old count may still be in flight
    for (auto &pd : passes_pending) pass_events_free.push_back(pd.ev);
    passes_pending.clear(); frames_completed_known = 0;
    frames_rendered = 0;
    return MRT_OK;
}

// ---- scene queries
int query_closest(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, MRTIntersection *out) {
    if (n == 0) return MRT_OK;
    DevBuf<MRTRay> d_r; DevBuf<MRTIntersection> d_o;
    MRT_HIP(d_r.alloc(n)); MRT_HIP(d_o.alloc(n));
    MRT_HIP(hipMemcpyAsync(d_r.p, rays, n * sizeof(MRTRay), hipMemcpyHostToDevice, stream));
    if (sc.num_wnodes) hipLaunchKernelGGL(k_query_closest<true>, dim3(cdiv(n, 64)), dim3(64), (size_t)sc.wide_depth * WIDE_STACK_LEVEL_BYTES, stream, sc.view(), d_r.p, (uint32_t)n, d_o.p);
    else hipLaunchKernelGGL(k_query_closest<false>, dim3(cdiv(n, 64)), dim3(64), 0, stream, sc.view(), d_r.p, (uint32_t)n, d_o.p);
    MRT_HIP(hipMemcpyAsync(out, d_o.p, n * sizeof(MRTIntersection), hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    MRT_HIP(hipGetLastError());
    return MRT_OK;
}
int query_any(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, int32_t *out) {
    if (n == 0) return MRT_OK;
    DevBuf<MRTRay> d_r; DevBuf<int32_t> d_o;
    MRT_HIP(d_r.alloc(n)); MRT_HIP(d_o.alloc(n));
    MRT_HIP(hipMemcpyAsync(d_r.p, rays, n * sizeof(MRTRay), hipMemcpyHostToDevice, stream));
    if (sc.num_wnodes) hipLaunchKernelGGL(k_query_any<true>, dim3(cdiv(n, 64)), dim3(64), (size_t)sc.wide_depth * WIDE_STACK_LEVEL_BYTES, stream, sc.view(), d_r.p, (uint32_t)n, d_o.p);
    else hipLaunchKernelGGL(k_query_any<false>, dim3(cdiv(n, 64)), dim3(64), 0, stream, sc.view(), d_r.p, (uint32_t)n, d_o.p);
    MRT_HIP(hipMemcpyAsync(out, d_o.p, n * 4, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    MRT_HIP(hipGetLastError());
    return MRT_OK;
}

int query_stats(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, int any, uint32_t *out4) {
    if (n == 0) return MRT_OK;
    DevBuf<MRTRay> d_r; DevBuf<uint32_t> d_o;
    MRT_HIP(d_r.alloc(n)); MRT_HIP(d_o.alloc(8 * n));
    MRT_HIP(hipMemcpyAsync(d_r.p, rays, n * sizeof(MRTRay), hipMemcpyHostToDevice, stream));
    if (sc.num_wnodes) hipLaunchKernelGGL(k_query_stats<true>, dim3(cdiv(n, 64)), dim3(64), (size_t)sc.wide_depth * WIDE_STACK_LEVEL_BYTES, stream, sc.view(), d_r.p, (uint32_t)n, any, d_o.p);
    else hipLaunchKernelGGL(k_query_stats<false>, dim3(cdiv(n, 64)), dim3(64), 0, stream, sc.view(), d_r.p, (uint32_t)n, any, d_o.p);
    MRT_HIP(hipMemcpyAsync(out4, d_o.p, n * 32, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    MRT_HIP(hipGetLastError());
    return MRT_OK;
}

int query_stream_stats(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, int any, uint32_t per_wave, uint32_t *out8, size_t nwaves) {
    if (n == 0 || sc.num_wnodes == 0) return MRT_OK;
    DevBuf<MRTRay> d_r; DevBuf<uint32_t> d_o;
    MRT_HIP(d_r.alloc(n)); MRT_HIP(d_o.alloc(8 * nwaves));
    MRT_HIP(hipMemsetAsync(d_o.p, 0, 32 * nwaves, stream));
    MRT_HIP(hipMemcpyAsync(d_r.p, rays, n * sizeof(MRTRay), hipMemcpyHostToDevice, stream));
    const size_t lds = (size_t)sc.wide_depth * WIDE_STACK_LEVEL_BYTES + (sc.num_inst ? WIDE_WORLD_RAY_BYTES : 0);
    if (sc.num_inst) hipLaunchKernelGGL(k_query_stream_stats<true>, dim3((uint32_t)nwaves), dim3(64), lds, stream, sc.view(), d_r.p, (uint32_t)n, any, per_wave, (uint32_t)sc.wide_depth, d_o.p);
    else hipLaunchKernelGGL(k_query_stream_stats<false>, dim3((uint32_t)nwaves), dim3(64), lds, stream, sc.view(), d_r.p, (uint32_t)n, any, per_wave, (uint32_t)sc.wide_depth, d_o.p);
    MRT_HIP(hipMemcpyAsync(out8, d_o.p, 32 * nwaves, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    MRT_HIP(hipGetLastError());
    return MRT_OK;
}

int query_stream(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, int any, MRTIntersection *out) {
    if (n == 0) return MRT_OK;
    if (sc.num_wnodes == 0) { set_error("the scene has no 8-wide layout"); return MRT_ERR_UNSUPPORTED; }
    for (size_t i = 0; i < n; i++) if (rays[i].min_distance != 0.0f) { set_error("the stream traversal starts every ray at distance 0 (min_distance must be 0)"); return MRT_ERR_INVALID_ARGUMENT; }
    if (n >= (size_t(1) << 31)) { set_error("too many rays"); return MRT_ERR_INVALID_ARGUMENT; }
    DevBuf<MRTRay> d_r; DevBuf<MRTIntersection> d_o;
    MRT_HIP(d_r.alloc(n)); MRT_HIP(d_o.alloc(n));
    MRT_HIP(hipMemcpyAsync(d_r.p, rays, n * sizeof(MRTRay), hipMemcpyHostToDevice, stream));
    const uint32_t per_wave = 256;
    const size_t lds = (size_t)sc.wide_depth * WIDE_STACK_LEVEL_BYTES + (sc.num_inst ? WIDE_WORLD_RAY_BYTES : 0);
    if (sc.num_inst) hipLaunchKernelGGL(k_query_stream<true>, dim3(cdiv(n, per_wave)), dim3(64), lds, stream, sc.view(), d_r.p, (uint32_t)n, any, per_wave, d_o.p);
    else hipLaunchKernelGGL(k_query_stream<false>, dim3(cdiv(n, per_wave)), dim3(64), lds, stream, sc.view(), d_r.p, (uint32_t)n, any, per_wave, d_o.p);
    MRT_HIP(hipMemcpyAsync(out, d_o.p, n * sizeof(MRTIntersection), hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    MRT_HIP(hipGetLastError());
    return check_bounds_record();
}

int probe_halton(hipStream_t stream, const int32_t *i, const int32_t *d, size_t n, float *out) {
    if (n == 0) return MRT_OK;
    DevBuf<int32_t> di, dd; DevBuf<float> dout;
    MRT_HIP(di.alloc(n)); MRT_HIP(dd.alloc(n)); MRT_HIP(dout.alloc(n));
    MRT_HIP(hipMemcpyAsync(di.p, i, n * 4, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(dd.p, d, n * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_probe_halton, dim3(cdiv(n, 256)), dim3(256), 0, stream, di.p, dd.p, (uint32_t)n, dout.p);
    MRT_HIP(hipMemcpyAsync(out, dout.p, n * 4, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    return MRT_OK;
}
int probe_hemisphere(hipStream_t stream, const float *u2, const float *n3, size_t n, float *out3) {
    if (n == 0) return MRT_OK;
    DevBuf<float> du, dn, dout;
    MRT_HIP(du.alloc(2 * n)); MRT_HIP(dn.alloc(3 * n)); MRT_HIP(dout.alloc(3 * n));
    MRT_HIP(hipMemcpyAsync(du.p, u2, n * 8, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(dn.p, n3, n * 12, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_probe_hemisphere, dim3(cdiv(n, 256)), dim3(256), 0, stream, du.p, dn.p, (uint32_t)n, dout.p);
    MRT_HIP(hipMemcpyAsync(out3, dout.p, n * 12, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    return MRT_OK;
}
int probe_seeds(hipStream_t stream, uint32_t seed, int w, int h, uint32_t *out) {
    size_t n = (size_t)w * h;
    if (n == 0) return MRT_OK;
    DevBuf<uint32_t> d; MRT_HIP(d.alloc(n));
    hipLaunchKernelGGL(k_seed, dim3(cdiv(n, 256)), dim3(256), 0, stream, d.p, (uint32_t)n, seed);
    MRT_HIP(hipMemcpyAsync(out, d.p, n * 4, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    return MRT_OK;
}

}  // namespace mrt
