// flow.h — one launch per pass: the bounces of a batch of frames as a dataflow of ray granules (renderer option flow = 1; EXPERIMENT, off by default).
// Included by renderer.hip inside namespace mrt { namespace { ... after FrameParams, primary_ray and the traversal headers.
//
// The wavefront pipeline (renderer.hip) runs a pass as eight dependent launches: trace_primary, then per bounce { shade, trace }.  Every
// launch ends in a tail and starts with a ramp, nothing of bounce b + 1 can start before the last wave of bounce b has left, and the chip is
// only kept full by running many passes side by side (12 streams x 4 frames).  Here the pass is ONE launch of resident waves after the
// primary trace (the previous review's item 2).  Work is handed from wave to wave in GRANULES of up to `granule` consecutive queue entries:
//
//   stage 0      a chunk of primary slots (hits and directions written by k_trace_primary)  -> shade(0)
//   stage b >= 1 granules of Q[b-1], the bounce rays shade(b-1) appended                     -> closest hit -> shade(b)
//   shadow b     granules of SQ[b], the shadow rays shade(b) appended                        -> any hit -> lit[b][pixel] = 1
//
// shade(b) writes the light's contribution into con[b][pixel]; k_accumulate_flow adds the contributions whose shadow ray got through, in bounce
// order — the additions of Raytracing.metal:371-373 in the same order on the same floats, so the image is the pipeline's bit for bit.
// (The pipeline's read-modify-write of one sample buffer would need the shadow rays of different bounces ordered against each other.)
//
// A wave runs SESSIONS.  Stage 0: a chunk of primary slots, shaded.  Otherwise a traversal session: traverse_wide_stream with a ray supply that
// pulls granules from the queues as lanes free up — bounce rays of one stage (the granules are remembered, one per lane) and, to fill up, shadow
// rays of any bounce — until nothing is ready or `session_rays` rays have been started; when the last ray is done the wave shades the hits of its
// bounce-ray granules, publishes what that appended, and looks for the next session (breadth first: the shallowest stage that has work).
// Nothing ready: the wave polls one word for a while, or leaves when little is outstanding.  That is safe: whoever publishes work looks for work
// afterwards, so nothing is left behind by a wave that has gone; and no wave ever waits for another one, so the launch cannot hang whatever the
// residency.  k_accumulate_flow checks that every published ray was consumed (Renderer::stats() reports a pass that ended with work left).
//
// Hand-off between waves (MI355X: eight XCDs with private L2s, MI355X_MICROARCH.md "inter-workgroup visibility").  A session reserves room for
// everything it can append with ONE packed atomic on the bounce's cursors (it knows how many of its rays hit: every hit appends a bounce ray
// and at most one shadow ray, so the shadow queue may keep a gap), stores the entries write-through (buffer_store_dwordx4 sc1), waits for the
// stores (s_waitcnt vmcnt(0)), and pushes descriptors {first entry, count <= granule} onto the queue's ready ring; when they have landed it
// releases as many permits.  A consumer takes up to K permits (fetch-add; what it took too many it hands back), then as many tickets for ring
// slots, and reads the descriptors there.  Fetch-adds only — a compare-and-swap cursor gave one claim per memory round trip with 6 000 waves
// competing, and one reservation per 64 rays ran three cursor words at their ~88 atomics per microsecond (both measured: a quarter and a half
// of the pipeline's rate).  Every load of a queue entry is an sc1 load.  A wave reads back its own hit records (plain stores, drained, sc1
// loads: same XCD).  con[] / lit[] are written once and read by the next launch.
//
// Measured (profiles/r03_flow.txt): the image is the pipeline's bit for bit and the instruction count per frame is the pipeline's
// (291 M against 294 M of k_shade + the traversal launches), but the fused kernel needs 128 registers to stay out of scratch (4 waves per SIMD;
// at the traversal's 80 it spills 56) and a pass's waves thin out towards its end; 12 x 4 frames: 0.72 ms per frame against the pipeline's 0.565,
// one frame alone 2.9 ms against 1.75.  Not the default.
#pragma once

typedef unsigned int flow_u32x4 __attribute__((ext_vector_type(4)));
#define FLOW_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

MRT_DEV __amdgpu_buffer_rsrc_t flow_rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)0xFFFFFFF0u, 0x00020000); }
MRT_DEV float4 flow_ld16(__amdgpu_buffer_rsrc_t r, uint32_t index) {       // 16-byte entry `index`, sc1: not from this CU's L1
    const flow_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, index * 16u, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
MRT_DEV void flow_st16(__amdgpu_buffer_rsrc_t r, uint32_t index, float4 a) {   // write-through
    flow_u32x4 v; v.x = __float_as_uint(a.x); v.y = __float_as_uint(a.y); v.z = __float_as_uint(a.z); v.w = __float_as_uint(a.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, index * 16u, 0, 16);
}
MRT_DEV uint32_t flow_uload(const uint32_t *p) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, FLOW_RLX_AGENT)); }   // wave-uniform address
// lane k's value of v, k wave-uniform: the result is a scalar.  (__shfl returns it in a vector register, every use after it counts as divergent — ranges, loop
// bounds, buffer descriptors chosen by them — and a build of this kernel with __shfl here traced a few hundred wrong rays per frame, deterministically, while a
// build with one more condition in the source was exact: ds_bpermute returns 0 for a lane the compiler has masked off.  Keep uniform values scalar.)
MRT_DEV uint32_t flow_lane(uint32_t v, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)k)); }
MRT_DEV void flow_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
MRT_DEV uint32_t flow_uadd(uint32_t *p, uint32_t v, uint32_t lane) {          // one returning atomic for the wave (wave-uniform address and value)
    uint32_t old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(p, v, FLOW_RLX_AGENT);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}

constexpr uint32_t FLOW_LINE = 32;          // every shared word on a 128-byte line of its own
constexpr int FLOW_MAX_BOUNCES = 3;
constexpr int FLOW_QUEUES = 5;              // queue id: 0, 1 = Q[0], Q[1] (bounce rays); 2 + b = SQ[b] (shadow rays)
constexpr uint32_t FLOW_MAX_TAKE = 16;      // descriptors per session
enum : uint32_t { FW_TAKEN_P = 0, FW_AVAIL = 1, FW_HEAD = 6, FW_TAIL = 11, FW_PENDING = 16, FW_TRUE_Q = 17, FW_TRUE_S = 20, FW_ERROR = 23, FW_STATS = 24 /* MRT_FLOW_STATS build: 16 words */, FW_WORDS = 40 };
constexpr uint32_t FLOW_HEADER_WORDS = FW_WORDS * FLOW_LINE;

// The pass's buffers, as a table of pointers in DEVICE memory (FrameLane::f_tab): the kernel reads the few entries a session needs when it needs them.
// (As a by-value kernel argument the thirty pointers were loaded ahead of the session loop and held in scalar registers throughout: 106 of them,
// spilled into vector registers that then spilled to scratch — 472 bytes per lane, and half the rate.)
enum : int {
    FT_QA = 0, FT_QB = 2,            // Q[b]: bounce rays appended by shade(b), traced by stage b + 1
    FT_SA = 4, FT_SB = 7,            // SQ[b]: shadow rays appended by shade(b)
    FT_HITS = 10,                    // [0]: the pass's primary hits (k_trace_primary), index = sub-frame * capacity + slot; [b >= 1]: closest hits of Q[b - 1], by queue position
    FT_CON = 13,                     // [b][sub-frame * npix + pixel]: the light's contribution at bounce b (Raytracing.metal:372)
    FT_LIT = 16,                     // [b]: byte b of pixel 0's word; [4 * (sub-frame * npix + pixel) + b] = 1: the shadow ray of bounce b was not occluded (:371)
    FT_RING = 19,                    // [q]: descriptors in the order they were published: first entry | count << 32 (count > 0); zero at launch
    FT_DIRS = 24,                    // primary directions | sample index (k_trace_primary)
    FT_COUNT = 25
};
struct FlowArgs {
    const void *const *tab;          // FT_* (device memory)
    unsigned long long *counts;      // [b] reservation cursors {lo: Q[b], hi: SQ[b]} (FrameLane::bounce_counts)
    uint32_t *words;                 // FW_* header, every word FLOW_LINE apart; zero at launch.  FW_PENDING: rays published and not yet traced / shaded, as a difference from n_primary
    uint32_t n_primary;              // capacity * batch
    uint32_t chunk;                  // primary slots per stage-0 session (multiple of 64)
    uint32_t take;                   // K: descriptors a claim takes at most (<= FLOW_MAX_TAKE)
    uint32_t session_rays;           // rays a traversal session may start before it drains and shades (<= 64 granules of bounce rays)
    uint32_t mix;                    // 1: a session that traces bounce rays fills up with shadow rays
    uint32_t granule;                // rays per descriptor (multiple of 64): what a session appended is published in pieces of this size
    uint32_t breadth_first;          // 1: the shallowest stage with work first (even stages, large queues); 0: the deepest (short queues, a long tail)
    uint32_t idle_polls;             // polls without work before a wave leaves whatever is outstanding
    uint32_t exit_rays;              // a wave without work leaves when fewer rays than this are outstanding in the pass
};
template <class T> MRT_DEV T *flow_ptr(const FlowArgs &fa, int entry) {          // wave-uniform entry: a scalar load
    return static_cast<T *>(const_cast<void *>(fa.tab[__builtin_amdgcn_readfirstlane(entry)]));
}

#ifdef MRT_FLOW_STATS      // diagnostics build: where does a wave's time go?  (ticks of the 100 MHz wall clock, summed over the launch's waves; printed by Renderer::wait())
#define FLOW_TICK(var) const unsigned long long var = wall_clock64()
#define FLOW_ADD(slot, dt) do { if (lane == 0) __hip_atomic_fetch_add(W + (FW_STATS + (slot)) * FLOW_LINE, (uint32_t)(dt), FLOW_RLX_AGENT); } while (0)
#else
#define FLOW_TICK(var) do {} while (0)
#define FLOW_ADD(slot, dt) do {} while (0)
#endif
// up to K ready descriptors of queue q: permits, tickets, ring slots.  Lane i < returned count holds descriptor i in (base, cnt).
MRT_DEV uint32_t flow_claim(const FlowArgs &fa, int q, uint32_t lane, uint32_t &base, uint32_t &cnt) {
    uint32_t *const W = fa.words;
    FLOW_TICK(tc0); FLOW_ADD(11, 1);
    const int avail = (int)flow_uload(W + (FW_AVAIL + q) * FLOW_LINE);
    if (avail <= 0) return 0u;
    const int K = (int)fa.take;
    const int old = (int)flow_uadd(W + (FW_AVAIL + q) * FLOW_LINE, (uint32_t)-K, lane);
    const int got = old >= K ? K : (old > 0 ? old : 0);
    if (got < K && lane == 0) __hip_atomic_fetch_add(W + (FW_AVAIL + q) * FLOW_LINE, (uint32_t)(K - got), FLOW_RLX_AGENT);      // hand back what was not there
    if (got == 0) { FLOW_TICK(tc2); FLOW_ADD(2, tc2 - tc0); return 0u; }
    const uint32_t ticket = flow_uadd(W + (FW_HEAD + q) * FLOW_LINE, (uint32_t)got, lane);
    unsigned long long *const ring = flow_ptr<unsigned long long>(fa, FT_RING + q);
    unsigned long long v = 1ull;
    if (lane < (uint32_t)got) {
        v = 0ull;
        for (uint32_t spin = 0; spin < (1u << 16) && v == 0ull; spin++) {        // a slot's pusher reserved it before it released the permit and has waited for the store: one look normally
            v = __hip_atomic_load(ring + ticket + lane, FLOW_RLX_AGENT);
            if (v == 0ull) __builtin_amdgcn_s_sleep(8);
        }
    }
    if (__ballot(v == 0ull) != 0ull) { if (lane == 0) __hip_atomic_fetch_add(W + FW_ERROR * FLOW_LINE, 1u, FLOW_RLX_AGENT); return 0u; }
    base = (uint32_t)v; cnt = (uint32_t)(v >> 32);
    FLOW_TICK(tc1); FLOW_ADD(0, tc1 - tc0); FLOW_ADD(15, got);
    return (uint32_t)got;
}
// The ray supply of a traversal session: granules are pulled from the pass's queues as the wave's lanes free up (traverse_wide_stream refills them), so the wave
// only runs dry when no queue has anything ready — or at its budget.  Bounce rays come from ONE stage's queue (their hits are shaded by this wave afterwards:
// the granules are remembered, one per lane), shadow rays from any.
struct FlowSession {
    int cstage;                          // the closest-hit stage this session pulls bounce rays for (>= 1), 0 = shadow rays only
    uint32_t budget;                     // rays the session may still start
    uint32_t c_base, c_cnt, c_n, c_k; int c_q;      // claimed and not yet handed out: lane i holds descriptor i; the queue they are from
    __amdgpu_buffer_rsrc_t rA, rB; int cur_q;       // the queue of the chunk being fetched
    uint32_t g_base, g_cnt, ng;          // the session's bounce-ray granules (or the primary slots of a stage-0 session): lane i holds granule i
    uint32_t n_closest, n_shadow;        // rays handed out
};
struct FlowSource {
    const FlowArgs &fa; FlowSession &st; uint32_t lane; int nb;
    MRT_DEV bool operator()(uint32_t &ob, uint32_t &oe) {
        for (;;) {
            if (st.c_k < st.c_n) {
                ob = flow_lane(st.c_base, st.c_k); const uint32_t cnt = flow_lane(st.c_cnt, st.c_k); oe = ob + cnt; st.c_k++;
                if (st.cur_q != st.c_q) {
                    st.cur_q = st.c_q;
                    st.rA = flow_rsrc(flow_ptr<void>(fa, st.c_q < 2 ? FT_QA + st.c_q : FT_SA + st.c_q - 2)); st.rB = flow_rsrc(flow_ptr<void>(fa, st.c_q < 2 ? FT_QB + st.c_q : FT_SB + st.c_q - 2));
                }
                if (st.c_q < 2) { if (lane == st.ng) { st.g_base = ob; st.g_cnt = cnt; } st.ng++; st.n_closest += cnt; } else st.n_shadow += cnt;
                st.budget -= min(st.budget, cnt);
                return true;
            }
            if (st.budget == 0u) return false;
            st.c_n = 0; st.c_k = 0;
            if (st.cstage >= 1 && st.ng + fa.take <= 64u) { st.c_n = flow_claim(fa, st.cstage - 1, lane, st.c_base, st.c_cnt); st.c_q = st.cstage - 1; }
#pragma unroll
            for (int b = 0; b < FLOW_MAX_BOUNCES; b++)
                if (st.c_n == 0u && b < nb && (fa.mix != 0u || st.n_closest == 0u)) { st.c_n = flow_claim(fa, 2 + b, lane, st.c_base, st.c_cnt); st.c_q = 2 + b; }
            if (st.c_n == 0u) return false;
        }
    }
};

#ifndef MRT_FLOW_WAVES
#define MRT_FLOW_WAVES 4      // waves per SIMD the kernel is compiled for: 128 registers, no scratch.  At 5 (96 registers) it spills 40, at 6 (80: what the traversal alone needs) 56, and is 15 % / 50 % slower
#endif
__global__ void __launch_bounds__(64, MRT_FLOW_WAVES) k_flow(SceneView s, FrameParams fp, FlowArgs fa, const uint32_t *__restrict__ seeds) {
    extern __shared__ uint32_t stk_dyn[];
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int nb = fp.max_bounces;
    uint32_t *const W = fa.words;
    bool p_left = true;
    uint32_t idle = 0;
    FLOW_TICK(t_born);
    for (;;) {
        FLOW_TICK(t0);
        FlowSession st{};
        st.cur_q = -1;
        int stage = -1;              // the stage whose hits this session shades
        auto take_primary = [&]() {
            const uint32_t base = flow_uadd(W + FW_TAKEN_P * FLOW_LINE, fa.chunk, lane);
            if (base < fa.n_primary) { stage = 0; st.ng = 1; st.g_base = base; st.g_cnt = min(fa.n_primary - base, fa.chunk); }
            else p_left = false;
        };
        // breadth first: primary slots while there are any, then rays — bounce rays of the shallowest stage that has some; deepest first: the other way round
        if (fa.breadth_first && p_left) take_primary();
        if (stage < 0) {
#pragma unroll
            for (int k = 1; k < FLOW_MAX_BOUNCES; k++) {
                const int b = fa.breadth_first ? k : FLOW_MAX_BOUNCES - k;
                if (b < nb && st.cstage == 0 && (int)flow_uload(W + (FW_AVAIL + b - 1) * FLOW_LINE) > 0) st.cstage = b;
            }
            st.budget = fa.session_rays;
            float4 *__restrict__ const hits_b = flow_ptr<float4>(fa, FT_HITS + st.cstage);
            uint8_t *__restrict__ const lit0 = flow_ptr<uint8_t>(fa, FT_LIT);
            traverse_wide_stream<false>(s, FlowSource{fa, st, lane, nb}, stk_dyn,
                [&](uint32_t i, float4 &A, float4 &B, uint32_t &tag, uint32_t &is_any) {
                    A = flow_ld16(st.rA, i); B = flow_ld16(st.rB, i);
                    if (st.cur_q >= 2) { tag = __float_as_uint(B.w) | ((uint32_t)(st.cur_q - 2) << 27); is_any = 1u; }        // a shadow ray reports to its pixel's byte of its bounce's plane
                    else { tag = i; is_any = 0u; A.w = __builtin_inff(); }        // a bounce ray's tmax word carries the throughput chain
                },
                [&](uint32_t j, bool is_any, bool hit, const TravHit &h) {
                    if (is_any) { if (!hit) lit0[4 * (size_t)(j & 0x7FFFFFFu) + (j >> 27)] = 1; }
                    else hits_b[j] = hit ? make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                });
            flow_drain();
            if (st.n_closest != 0u) stage = st.cstage;
            FLOW_TICK(t2); FLOW_ADD(1, t2 - t0); FLOW_ADD(8, st.n_closest); FLOW_ADD(9, st.n_shadow); FLOW_ADD(10, (st.n_closest + st.n_shadow) != 0u ? 1 : 0);
        }
        if (stage < 0 && st.n_shadow == 0u && !fa.breadth_first && p_left) take_primary();
        if (stage < 0) {
            if (st.n_shadow != 0u) { if (lane == 0) __hip_atomic_fetch_add(W + FW_PENDING * FLOW_LINE, 0u - st.n_shadow, FLOW_RLX_AGENT); idle = 0; continue; }
            // nothing ready.  The wave waits (polling one word, with growing pauses) while there are rays in flight whose sessions will publish more work — but not
            // for the crumbs: once fewer than `exit_rays` rays are outstanding the waves still in a session take care of them (whoever publishes work looks for
            // work afterwards, so nothing is ever left behind), and this wave's slot is worth more to the next pass's launch
            const uint32_t outstanding = flow_uload(W + FW_PENDING * FLOW_LINE) + fa.n_primary;
            if (outstanding < fa.exit_rays || ++idle > fa.idle_polls) break;
            for (uint32_t k = 0; k < min(idle, 8u); k++) __builtin_amdgcn_s_sleep(127);
            FLOW_TICK(t_idle); FLOW_ADD(4, t_idle - t0); FLOW_ADD(5, 1);
            continue;
        }
        idle = 0;
        const uint32_t nd = st.ng, d_base = st.g_base, d_cnt = st.g_cnt;
        uint32_t n_session = 0;
        for (uint32_t k = 0; k < nd; k++) n_session += flow_lane(d_cnt, k);
        const uint32_t n_traced_shadow = st.n_shadow;
        FLOW_TICK(t3);
        // ---- shade the session's hits (Raytracing.metal:249-391, the diffuse path of k_shade<false, false, true>), 64 at a time
        const int bounce = stage;
        const bool first = stage == 0;
        const __amdgpu_buffer_rsrc_t rqA = flow_rsrc(flow_ptr<void>(fa, first ? FT_DIRS : FT_QA + stage - 1)), rqB = flow_rsrc(flow_ptr<void>(fa, first ? FT_DIRS : FT_QB + stage - 1)), rH = flow_rsrc(flow_ptr<void>(fa, FT_HITS + stage));
        const bool has_next = bounce + 1 < nb;
        const __amdgpu_buffer_rsrc_t rnA = flow_rsrc(flow_ptr<void>(fa, has_next ? FT_QA + bounce : FT_DIRS)), rnB = flow_rsrc(flow_ptr<void>(fa, has_next ? FT_QB + bounce : FT_DIRS));      // (no next bounce: never stored to)
        const __amdgpu_buffer_rsrc_t rsA = flow_rsrc(flow_ptr<void>(fa, FT_SA + bounce)), rsB = flow_rsrc(flow_ptr<void>(fa, FT_SB + bounce));
        float4 *__restrict__ const con_b = flow_ptr<float4>(fa, FT_CON + bounce);
        // room for everything the session can append: every hit appends a bounce ray (when there is a next bounce) and at most one shadow ray
        uint32_t n_hits = 0;
        for (uint32_t k = 0; k < nd; k++) {
            const uint32_t rb = flow_lane(d_base, k), re = rb + flow_lane(d_cnt, k);
            for (uint32_t base = rb; base < re; base += 64u) {
                const uint32_t i = base + lane;
                const bool hit = i < re && __builtin_amdgcn_raw_buffer_load_b32(rH, i * 16u + 12u, 0, 16) != 0xFFFFFFFFu;
                n_hits += (uint32_t)__popcll(__ballot(hit));
            }
        }
        uint32_t cur_n = 0, cur_s = 0;
        if (n_hits != 0u) {
            unsigned long long got = 0;
            if (lane == 0) got = __hip_atomic_fetch_add(fa.counts + bounce, ((unsigned long long)n_hits << 32) | (has_next ? n_hits : 0u), FLOW_RLX_AGENT);
            cur_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got); cur_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32));
        }
        const uint32_t first_n = cur_n, first_s = cur_s;
        for (uint32_t k = 0; k < nd; k++) {
        const uint32_t rb = flow_lane(d_base, k), re = rb + flow_lane(d_cnt, k);
        for (uint32_t base = rb; base < re; base += 64u) {
            const uint32_t i = base + lane;
            bool active = i < re;
            float4 A = make_float4(0, 0, 0, 0), B = A, H = make_float4(-1, 0, 0, __uint_as_float(0xFFFFFFFFu));
            if (active) {
                H = flow_ld16(rH, i);
                if (first) { B = flow_ld16(rqB, i); A = make_float4(fp.cam_pos.x, fp.cam_pos.y, fp.cam_pos.z, 0.0f); }     // :214 (rqB: the primary directions)
                else { A = flow_ld16(rqA, i); B = flow_ld16(rqB, i); }
            }
            const uint32_t gid = __float_as_uint(H.w);
            active = active && gid != 0xFFFFFFFFu;                             // :246-247 miss terminates the path
            bool want_shadow = false, want_next = false;
            f3 nrm = mk3(0, 1, 0), ldir = mk3(0, 1, 0), lcol = mk3(0, 0, 0), color = mk3(0, 0, 0), ndir = mk3(0, 1, 0), norg = mk3(0, 0, 0);
            float ldist = 0.0f; uint32_t pix = 0, chain_out = 0;
            if (active) {
                float4 Cc;
                if (first) Cc = make_float4(1.0f, 1.0f, 1.0f, 0.0f);           // :226
                else {
                    const uint32_t ch = __float_as_uint(A.w);                  // resource slots of bounce 0 (low half) and, at bounce 2, of bounce 1 (high half)
                    Cc = s.base_color[ch & 0xFFFFu];
                    if (bounce >= 2) { const float4 s1 = s.base_color[ch >> 16]; Cc = make_float4(Cc.x * s1.x, Cc.y * s1.y, Cc.z * s1.z, 0.0f); }
                    chain_out = ch;
                }
                pix = __float_as_uint(B.w);
                const uint4 ts = s.tri_shade[gid];
                const uint32_t inst = ts.w >> 16, geom = ts.w & 0xFFFFu;
                const float bu = H.y, bv = H.z;
                const f3 P = mk3(A) + mk3(B) * H.x;                            // :261
                const float bw = 1.0f - bu - bv;                               // :63-64
                const f3 n_obj = (bu * mk3(s.normals[ts.y]) + bv * mk3(s.normals[ts.z])) + bw * mk3(s.normals[ts.x]);   // :66-72
                const f3 c0 = mk3(s.inst_cols[inst * 4 + 0]), c1 = mk3(s.inst_cols[inst * 4 + 1]), c2 = mk3(s.inst_cols[inst * 4 + 2]);
                const f3 n_w = mk3((c0.x * n_obj.x + c1.x * n_obj.y) + c2.x * n_obj.z,
                                   (c0.y * n_obj.x + c1.y * n_obj.y) + c2.y * n_obj.z,
                                   (c0.z * n_obj.x + c1.z * n_obj.y) + c2.z * n_obj.z);   // :267
                nrm = normalize3(n_w);                                         // :268
                const uint32_t rslot = inst * (uint32_t)s.max_sub + geom;
                const f3 surf = mk3(s.base_color[rslot]);                      // :262-269
                chain_out = bounce == 0 ? rslot : (chain_out & 0xFFFFu) | (rslot << 16);
                const int idx = (int)(q2load(&seeds[pix]) + fp.sampleIndex);   // pix = sub * npix + pixel: the table entry already holds + sub
                const int dim0 = 2 + bounce * 5;
                norg = P + nrm * 1e-3f;                                        // :350, :390
                const float ls = halton_dev(idx, dim0 + 0);                    // :272
                const int li = min((int)(ls * (float)fp.lightCount), fp.lightCount - 1);   // :273
                const LightDev L = s.lights[li];
                const int ltype = __float_as_int(L.position.w);
                if (ltype == MRTLightTypeAreaLight) {                          // :281-290, :94-128
                    const float ax = halton_dev(idx, dim0 + 1) * 2.0f - 1.0f;
                    const float ay = halton_dev(idx, dim0 + 2) * 2.0f - 1.0f;
                    const f3 sp = (mk3(L.position) + mk3(L.right) * ax) + mk3(L.up) * ay;
                    ldir = sp - P;
                    ldist = length3(ldir);
                    const float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                    ldir = ldir * inv;
                    lcol = mk3(L.color) * (inv * inv);
                    lcol = lcol * saturatef(dot3(neg3(ldir), mk3(L.forward)));
                } else if (ltype == MRTLightTypeSpotlight) {                   // :292-316
                    ldir = mk3(L.position) - P;
                    ldist = length3(ldir);
                    const float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                    ldir = ldir * inv;
                    lcol = mk3(0, 0, 0);
                    const float spot = dot3(neg3(ldir), mk3(L.dirn));
                    if (spot > L.dirn.w) lcol = (mk3(L.color) * inv) * inv;
                } else if (ltype == MRTLightTypePointlight) {                  // :317-322
                    ldir = mk3(L.position) - P;
                    ldist = length3(ldir);
                    const float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                    ldir = ldir * inv;
                    lcol = (mk3(L.color) * inv) * inv;
                } else {                                                       // :323-327
                    ldir = neg3(mk3(L.dirn));
                    ldist = __builtin_inff();
                    lcol = mk3(L.color);
                }
                lcol = lcol * saturatef(dot3(nrm, ldir));                      // :331
                lcol = lcol * (float)fp.lightCount;                            // :335
                color = mk3(Cc) * surf;                                        // :339
                want_shadow = length3(lcol) > 0.0001f;                         // :341
                want_next = has_next;
                if (want_next) {
                    const float hx = halton_dev(idx, dim0 + 3), hy = halton_dev(idx, dim0 + 4);   // :384-385
                    ndir = align_hemisphere_dev(sample_cosine_hemisphere_dev(hx, hy), nrm);       // :387-388
                }
            }
            // append inside the session's reservation
            const unsigned long long m_sh = __ballot(want_shadow), m_nx = __ballot(want_next);
            if (want_shadow) {
                const uint32_t ss = cur_s + (uint32_t)__popcll(m_sh & lt);
                const f3 con = lcol * color;                                   // :372
                flow_st16(rsA, ss, make_float4(norg.x, norg.y, norg.z, ldist - 1e-3f));      // :350, :356
                flow_st16(rsB, ss, make_float4(ldir.x, ldir.y, ldir.z, __uint_as_float(pix)));
                con_b[pix] = make_float4(con.x, con.y, con.z, 0.0f);
            }
            if (want_next) {
                const uint32_t ns = cur_n + (uint32_t)__popcll(m_nx & lt);
                flow_st16(rnA, ns, make_float4(norg.x, norg.y, norg.z, __uint_as_float(chain_out)));      // :390 (tmax = inf: the traversal sets it)
                flow_st16(rnB, ns, make_float4(ndir.x, ndir.y, ndir.z, __uint_as_float(pix)));            // :391
            }
            cur_n += (uint32_t)__popcll(m_nx); cur_s += (uint32_t)__popcll(m_sh);
        }
        }
        // ---- publish: the stores have landed -> one descriptor per queue -> (landed) -> permits and the ray counts
        flow_drain();
        FLOW_TICK(t4); FLOW_ADD(3, t4 - t3); FLOW_ADD(12, n_session); FLOW_ADD(13, 1);
        const uint32_t n_next = cur_n - first_n, n_shadow = cur_s - first_s;
        {
            const int qn = bounce < 2 ? bounce : 0, qs = 2 + bounce;          // (the last bounce appends no bounce rays: n_next == 0)
            const uint32_t G = fa.granule, mn = (n_next + G - 1u) / G, ms = (n_shadow + G - 1u) / G;      // descriptors (<= 64: a session is at most 64 granules)
            if (mn) {
                const uint32_t pos = flow_uadd(W + (FW_TAIL + qn) * FLOW_LINE, mn, lane);
                if (lane < mn) __hip_atomic_store(flow_ptr<unsigned long long>(fa, FT_RING + qn) + pos + lane, (unsigned long long)(first_n + lane * G) | ((unsigned long long)min(G, n_next - lane * G) << 32), FLOW_RLX_AGENT);
            }
            if (ms) {
                const uint32_t pos = flow_uadd(W + (FW_TAIL + qs) * FLOW_LINE, ms, lane);
                if (lane < ms) __hip_atomic_store(flow_ptr<unsigned long long>(fa, FT_RING + qs) + pos + lane, (unsigned long long)(first_s + lane * G) | ((unsigned long long)min(G, n_shadow - lane * G) << 32), FLOW_RLX_AGENT);
            }
            flow_drain();
            if (lane == 0) {
                if (mn) { __hip_atomic_fetch_add(W + (FW_AVAIL + qn) * FLOW_LINE, mn, FLOW_RLX_AGENT); __hip_atomic_fetch_add(W + (FW_TRUE_Q + bounce) * FLOW_LINE, n_next, FLOW_RLX_AGENT); }
                if (ms) { __hip_atomic_fetch_add(W + (FW_AVAIL + qs) * FLOW_LINE, ms, FLOW_RLX_AGENT); __hip_atomic_fetch_add(W + (FW_TRUE_S + bounce) * FLOW_LINE, n_shadow, FLOW_RLX_AGENT); }
                __hip_atomic_fetch_add(W + FW_PENDING * FLOW_LINE, n_next + n_shadow - n_session - n_traced_shadow, FLOW_RLX_AGENT);
            }
        }
        FLOW_TICK(t5); FLOW_ADD(6, t5 - t4);
    }
    FLOW_TICK(t_died); FLOW_ADD(7, t_died - t_born); FLOW_ADD(14, 1);
}

// The flow pass's accumulate: the pixel's sample is the sum of the contributions whose shadow ray got through, in bounce order (the additions of
// Raytracing.metal:371-373 as the pipeline's shadow launches make them: 0 + c0, + c1, + c2), then the running average (:394-403).  Block 0 folds the
// queue cursors into the ray totals and clears them for the lane's next pass.
__global__ void __launch_bounds__(64) k_accumulate_flow(FrameParams fp, FlowArgs fa, const float4 *__restrict__ prev, float4 *__restrict__ dst, unsigned long long *__restrict__ totals, uint32_t primary) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long closest = primary, shadow = 0;
        for (int b = 0; b < fp.max_bounces; b++) {
            if (b + 1 < fp.max_bounces) closest += fa.words[(FW_TRUE_Q + b) * FLOW_LINE];
            shadow += fa.words[(FW_TRUE_S + b) * FLOW_LINE];
            fa.counts[b] = 0;
        }
        totals[0] += closest; totals[1] += shadow; totals[2] += primary;
        // every published ray was traced and shaded: the launch ended because the work was done, not because its waves left
        if (fa.words[FW_PENDING * FLOW_LINE] + fa.n_primary != 0u || fa.words[FW_ERROR * FLOW_LINE] != 0u) totals[3] += 1;      // reported by Renderer::stats()
    }
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    int x, y;
    if (!slot_to_pixel(fp, slot, x, y)) return;
    const uint32_t pix = (uint32_t)y * (uint32_t)fp.width + (uint32_t)x;
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int sub = 0; sub < fp.batch; sub++) {                           // the batch's frames, in frame order
        const size_t sp = (size_t)sub * fp.capacity + slot;
        float4 sm = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int b = 0; b < fp.max_bounces; b++)
            if (static_cast<const uint8_t *>(fa.tab[FT_LIT])[4 * sp + b]) { const float4 cc = qload(&static_cast<const float4 *>(fa.tab[FT_CON + b])[sp]); sm = make_float4(sm.x + cc.x, sm.y + cc.y, sm.z + cc.z, 0.0f); }
        const uint32_t frame = fp.frameIndex + (uint32_t)sub;
        if (frame > 0) {
            const float4 p = sub == 0 ? q2load(&prev[pix]) : c;
            const float fi = (float)frame, den = (float)(frame + 1);
            c.x = (sm.x + p.x * fi) / den; c.y = (sm.y + p.y * fi) / den; c.z = (sm.z + p.z * fi) / den;
        } else c = sm;
    }
    q2store(&dst[pix], make_float4(c.x, c.y, c.z, 1.0f));
}
