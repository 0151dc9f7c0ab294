// shade.h — the shading step of the wavefront pipeline (Raytracing.metal:249-391) and what a frame's kernels share:
// FrameParams (the Uniforms block + shard + pass), the slot -> pixel map, primary-ray generation (:171-221).
//
//   shade_entry        one queue entry: normal interpolation, light pick + evaluation, throughput, NEE shadow-ray emit,
//                      cosine-hemisphere bounce, wave-ballot compaction of both output queues            (:249-391)
//   k_shade_primary    bounce 0 with the primary ray generated and traced in the same launch              (:171-247 + the above)
//   k_shade_pack       bounces >= 1: the HITS of the bounce queue compacted in LDS, shaded on full waves
//   k_shade            one queue entry per thread (bounce 0 from hit records; A/B of the packed form; materials at bounce 0)
#pragma once
#include "device_math.h"
#include "traverse.h"
#include "traverse_wide.h"
#include "traverse_instanced.h"

namespace mrt {
namespace {

struct FrameParams {            // Uniforms (ShaderTypes.h:89-97) + shard + bounce
    int32_t width, height;
    uint32_t frameIndex;        // accumulation weight (Raytracing.metal:395-401)
    uint32_t sampleIndex;       // Halton index = seed offset + sampleIndex (:202); == frameIndex unless sample-sharded
    int32_t lightCount;
    float4 cam_pos, cam_right, cam_up, cam_fwd;
    int32_t shard_rank, shard_world;
    int32_t tiles_x, tiles_local;
    int32_t bounce, max_bounces;
    // frame batching: one pass of the pipeline carries `batch` consecutive frames.  Sub-frame s uses Halton index
    // sampleIndex + s (the offset is baked into its copy of the seed table), slots [s * capacity, (s + 1) * capacity)
    // of the primary queue, of the seed table and of the sample buffer / contribution planes (a SAMPLE INDEX =
    // s * capacity + slot: per-pixel state of a pass is laid out by the shard's own slots, so a rank of N holds 1/N of it
    // and a wave's 8x8 tile is 64 consecutive entries); k_accumulate applies the sub-frames in order.
    uint32_t npix, capacity;
    int32_t batch;
    // throughput chain (fused pipeline, diffuse-only, max_bounces <= 3, <= 65 536 resource slots): a bounce ray carries the
    // resource slots of the surfaces its path has left (16 bits each, in the tmax word — always +inf for a bounce ray, and
    // every traversal kernel takes it as such) instead of a 16-byte throughput record; the next shade multiplies the same
    // base colours in the same order (Raytracing.metal:339), so the floats are the ones the record would have held
    int32_t chain;
    // k_shade_pack: queue entries per workgroup (a multiple of the workgroup size; the host sizes it by the launch:
    // SHADE_PACK_RANGE for large queues, less for small ones so that the grid still fills the chip)
    uint32_t pack_range;
    // k_shade_primary on the 8-wide layout: 32-bit words of LDS stack per wave (the scene's wide-tree depth x WIDE_STACK_LEVEL_BYTES / 4)
    uint32_t wide_stack_words;
    // bounce 0 of a pass of several frames (renderer option frame_bundle): a wave takes bundle_per_wave slots x bundle_w
    // sub-frames — the rays of one pixel side by side, bundle_w = batch / ceil(batch / 8) rounded up of them (8 for passes
    // of 8, 16, 32; 7 for 7: nine bundles in a wave and one idle lane) — instead of the 64 slots of one tile in one
    // sub-frame; the launch is then one grid row of ceil(capacity x bundle_groups / bundle_per_wave) waves.  0 = off
    int32_t frame_bundle;
    uint32_t bundle_w, bundle_groups, bundle_per_wave, bundle_magic;      // bundle_magic = ceil(65536 / bundle_w): lane / bundle_w = lane * bundle_magic >> 16 for lane < 64
    // Halton values of bounce 0 from a table (Renderer::halton_tab): halton_tab[(d - 1) * halton_n + (i - halton_w0)] =
    // halton_dev(i, d) for the dimensions d = 1 .. 6 of the pixel jitter's second component, the light pick, the area
    // light's point and the first hemisphere sample (Raytracing.metal:203, :272, :284-285, :384-385), filled by halton_dev
    // itself.  The recurrence is ~18 % of the VALU cycles of k_shade_primary (up to 13 digits per value, a quarter-rate
    // 32-bit multiply per pair of digits); a bundled wave reads the values of a pixel's sub-frames — consecutive indices —
    // from one or two cache lines instead.  nullptr (passes of one frame, narrow bundles, an index outside the window):
    // the recurrence.
    const float *halton_tab; uint32_t halton_w0, halton_n;
};
#ifndef MRT_HALTON_INTERLEAVED
#define MRT_HALTON_INTERLEAVED 1      // the six dimensions of an index side by side (24 B): the eight sub-frames of a pixel read 192 contiguous bytes — three or four 64-byte requests — instead of one request in each of six planes (profiles/r06_shade_primary_bytes.txt: the table was 0.83 GB of k_shade_primary's fetches per launch)
#endif
constexpr uint32_t HALTON_TAB_DIMS = 6, HALTON_TAB_SPAN = (1u << 20) + (1u << 16);      // seed offsets are below 2^20 (Renderer.swift:259): the window serves 2^16 frames before it moves
MRT_DEV float halton_b0(const FrameParams &fp, int idx, int d /* 1 .. 6 */) {
    const uint32_t j = (uint32_t)idx - fp.halton_w0;
    if (fp.halton_tab != nullptr && j < fp.halton_n) return fp.halton_tab[MRT_HALTON_INTERLEAVED ? (size_t)j * HALTON_TAB_DIMS + (uint32_t)(d - 1) : (size_t)(d - 1) * fp.halton_n + j];
    return halton_dev(idx, d);
}

// local slot -> pixel: one wave = one 8x8 tile (Renderer.swift:295-300), tiles dealt round-robin to shards
MRT_DEV bool slot_to_pixel(const FrameParams &fp, uint32_t slot, int &x, int &y) {
    uint32_t lt = slot >> 6, k = slot & 63;
    if ((int)lt >= fp.tiles_local) return false;
    uint32_t tile = lt * (uint32_t)fp.shard_world + (uint32_t)fp.shard_rank;
    uint32_t ty = tile / (uint32_t)fp.tiles_x, tx = tile - ty * (uint32_t)fp.tiles_x;
    x = (int)(tx * 8 + (k & 7)); y = (int)(ty * 8 + (k >> 3));
    return x < fp.width && y < fp.height;
}

// ------------------------------------------------------------------ primary rays
// Raytracing.metal:175, :202-221
MRT_DEV void primary_ray(const FrameParams &fp, const uint32_t *__restrict__ seeds, uint32_t sample_index, int x, int y, f3 &org, f3 &dir) {
    uint32_t offset = q2load(&seeds[sample_index]);                      // :175 (+ sub-frame index)
    int idx = (int)(offset + fp.sampleIndex);
    float r0, r1;                                                        // :202-203
    r0 = halton_dev(idx, 0); r1 = halton_b0(fp, idx, 1);
    float px = (float)x + r0, py = (float)y + r1;                        // :204
    float uvx = px / (float)fp.width, uvy = py / (float)fp.height;       // :207
    uvx = uvx * 2.0f - 1.0f; uvy = uvy * 2.0f - 1.0f;                    // :208
    dir = normalize3((uvx * mk3(fp.cam_right) + uvy * mk3(fp.cam_up)) + mk3(fp.cam_fwd));   // :216-218
    org = mk3(fp.cam_pos);                                               // :214
}

// ------------------------------------------------------------------ shade
// Queue compaction.  Lanes ballot, waves post their two counts to LDS, and ONE packed 64-bit atomic per
// workgroup reserves the output ranges of both queues ({next rays: low word, shadow rays: high word}):
// a single counter word sustains only ~88 returning atomics/us on gfx950 (MI355X_MICROARCH.md, row
// "dequeue"), so per-wave atomics on 32 K waves would cost more than the shading itself.
#ifndef MRT_SHADE_THREADS
#define MRT_SHADE_THREADS 256
#endif
#ifndef MRT_SHADE_WAVES
#define MRT_SHADE_WAVES 6     // waves per SIMD the shade kernels are compiled for: they need 76-78 registers; capped at 72 (7 waves) they spill 16-48 bytes and the frame is 4 % slower, at 64 (8 waves) 6 % slower
#endif
#ifndef MRT_SHADE_WIDE_WAVES
#define MRT_SHADE_WIDE_WAVES 5     // k_shade_primary on the 8-wide layout: the walk holds a node (20 registers) and a packet (10) on top of the shading state
#endif
constexpr int SHADE_THREADS = MRT_SHADE_THREADS;
constexpr int SHADE_WAVES = SHADE_THREADS / 64;
#ifndef MRT_SHADE_PACK_RANGE
#define MRT_SHADE_PACK_RANGE 4096
#endif
constexpr uint32_t SHADE_PACK_RANGE = MRT_SHADE_PACK_RANGE;      // k_shade_pack: queue entries per workgroup, at most (FrameParams::pack_range)

// What a shade launch reads and writes.  The kernels take these as individual __restrict__ parameters (SHADE_IO_PARAMS) and gather them here:
// the no-alias guarantee of a kernel parameter follows the pointer through the struct, that of a struct member handed over by value does not —
// k_shade_pack then spills 12-24 bytes per lane.
struct ShadeIO {
    const uint32_t *__restrict__ seeds;
    const float4 *__restrict__ rayA, *__restrict__ rayB, *__restrict__ thr;      // the queue shade(b - 1) wrote (bounce 0 reads none: it regenerates the primary ray)
    const float4 *__restrict__ hits;                 // hit records of that queue — or, PAIRS, 64-bit keys
    const unsigned long long *__restrict__ count_in; // nullptr at bounce 0: `capacity` slots
    uint32_t capacity;
    float4 *__restrict__ nrayA, *__restrict__ nrayB, *__restrict__ nthr;        // out: the next bounce rays
    float4 *__restrict__ srayA, *__restrict__ srayB, *__restrict__ scon;        // out: shadow rays; PLANES: scon is this bounce's contribution plane, by sample index
    unsigned long long *__restrict__ count_out;      // lo = next rays, hi = shadow rays
    float4 *__restrict__ sample_primary;             // bounce 0: the pass's sample buffer (zeroed here unless PLANES)
    float4 *__restrict__ sample;                     // MATERIALS: emitted radiance is added here
    uint32_t *__restrict__ hint;                     // k_shade_primary: per pixel, the packet its primary ray hit last (or nullptr)
};
#define SHADE_IO_PARAMS const uint32_t *__restrict__ seeds, const float4 *__restrict__ rayA, const float4 *__restrict__ rayB, const float4 *__restrict__ thr, \
                        const float4 *__restrict__ hits, const unsigned long long *__restrict__ count_in, uint32_t capacity,                               \
                        float4 *__restrict__ nrayA, float4 *__restrict__ nrayB, float4 *__restrict__ nthr,                                                    \
                        float4 *__restrict__ srayA, float4 *__restrict__ srayB, float4 *__restrict__ scon, unsigned long long *__restrict__ count_out,       \
                        float4 *__restrict__ sample_primary, float4 *__restrict__ sample, uint32_t *__restrict__ hint
#define SHADE_IO_GATHER ShadeIO{seeds, rayA, rayB, thr, hits, count_in, capacity, nrayA, nrayB, nthr, srayA, srayB, scon, count_out, sample_primary, sample, hint}
struct ShadeShared { uint32_t w_next[SHADE_WAVES], w_shadow[SHADE_WAVES], w_spec[SHADE_WAVES]; unsigned long long blk_base; };

// PAIRS (two-level scenes, bounces >= 1, renderer option tl_pairs): a 64-bit key {t bits, global triangle id} left by the
// TLAS / BLAS passes becomes a hit record — the barycentrics come from re-testing the winning triangle in its instance's
// object space (the traversal's own test of the winner: the same U, V, |det|)
MRT_DEV float4 pairs_hit(const SceneView &s, const ShadeIO &io, const uint32_t i, const unsigned long long key) {
    if (key == ~0ull) return make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
    const uint32_t g = (uint32_t)key;
    const InstanceDev &I = s.inst[instance_of_gid(s, g)];
    const uint32_t pk = s.tri_packet[I.ts_base + (g - I.gid_base)];
    const float4 Aw = qload(&io.rayA[i]), Bw = qload(&io.rayB[i]);
    const float4 *__restrict__ q = s.wpackets + WPK * (size_t)pk;
    float t_, U, V, ad;
    (void)tri_test(q[0], q[1], q[2], to_object_point(I, mk3(Aw)), to_object_dir(I, mk3(Bw)), 0.0f, __builtin_inff(), t_, U, V, ad);
    return make_float4(__uint_as_float((uint32_t)(key >> 32)), U / ad, V / ad, __uint_as_float(g));
}

// Everything after the hit record: one entry per thread (i, its hit H, whether there is one) — called once, or per round of
// a packing workgroup.  Every thread of the workgroup calls it (two barriers inside).
//   MATERIALS  the materials extension (renderer option materials = 1)
//   CHAIN      FrameParams::chain (compile-time: the flag as a run-time branch cost 40 bytes of spills)
//   PLANES     the light's contribution goes to scon[sample index] of this bounce's plane instead of the shadow queue, and
//              nothing is zeroed (shadow planes, Renderer::shadow_planes)
//   B0TAB      bounce 0 inside k_shade_primary: dimensions 2 .. 6 from the Halton table when there is one; `Bprim` is the
//              primary ray's {direction | sample index}, still in registers
template <bool MATERIALS, bool CHAIN, bool PLANES, bool B0TAB>
MRT_DEV void shade_entry(const SceneView &s, const FrameParams &fp, const ShadeIO &io, ShadeShared &sh, const uint32_t i, const float4 H, bool active, const float4 Bprim) {
    uint32_t gid = __float_as_uint(H.w);
    active = active && gid != 0xFFFFFFFFu;                               // :246-247 miss terminates the path
    bool want_shadow = false, want_next = false;
    f3 P = mk3(0, 0, 0), nrm = mk3(0, 1, 0), ldir = mk3(0, 1, 0), lcol = mk3(0, 0, 0), color = mk3(0, 0, 0), ndir = mk3(0, 1, 0), norg = mk3(0, 0, 0);
    float ldist = 0.0f; uint32_t pix = 0;
    bool special = false;                // next ray comes from a specular / dielectric lobe (materials extension): queued behind the diffuse ones
    uint32_t chain_in = 0, chain_out = 0;   // throughput chain (FrameParams::chain)
    if (active) {
        float4 A, B, C;
        if (io.sample_primary) {
            A = make_float4(fp.cam_pos.x, fp.cam_pos.y, fp.cam_pos.z, __builtin_inff());   // :214
            B = B0TAB ? Bprim : qload(&io.rayB[i]);                      // direction | sample index, written by the primary trace
            C = make_float4(1.0f, 1.0f, 1.0f, 0.0f);                     // :226
        } else if (CHAIN) {
            A = qload(&io.rayA[i]); B = qload(&io.rayB[i]);
            const uint32_t ch = __float_as_uint(A.w);          // slots of bounce 0 (low half) and, at bounce 2, of bounce 1 (high half)
            chain_in = ch;
            C = s.base_color[ch & 0xFFFFu];                     // (1, 1, 1) * surf0 == surf0
            if (fp.bounce >= 2) { const float4 s1 = s.base_color[ch >> 16]; C = make_float4(C.x * s1.x, C.y * s1.y, C.z * s1.z, 0.0f); }
            A.w = __builtin_inff();
        } else { A = qload(&io.rayA[i]); B = qload(&io.rayB[i]); C = qload(&io.thr[i]); }
        pix = __float_as_uint(B.w);
        uint32_t inst, geom, rec = gid, vb = 0;
        if (s.num_inst) {           // two-level scene: the shading record belongs to the BLAS, the instance is found from the global triangle id
            inst = instance_of_gid(s, gid);
            const InstanceDev &I = s.inst[inst];
            rec = I.ts_base + (gid - I.gid_base); vb = I.vbase;
        } else inst = 0;
        float bu = H.y, bv = H.z;
        P = mk3(A) + mk3(B) * H.x;                                       // :261
        float bw = 1.0f - bu - bv;                                       // :63-64
        const uint4 ts = s.tri_shade[rec];
        if (!s.num_inst) inst = ts.w >> 16;
        geom = ts.w & 0xFFFFu;
        const f3 n_obj = (bu * mk3(s.normals[vb + ts.y]) + bv * mk3(s.normals[vb + ts.z])) + bw * mk3(s.normals[vb + ts.x]);   // :66-72
        f3 c0 = mk3(s.inst_cols[inst * 4 + 0]), c1 = mk3(s.inst_cols[inst * 4 + 1]), c2 = mk3(s.inst_cols[inst * 4 + 2]);
        f3 n_w = mk3((c0.x * n_obj.x + c1.x * n_obj.y) + c2.x * n_obj.z,
                     (c0.y * n_obj.x + c1.y * n_obj.y) + c2.y * n_obj.z,
                     (c0.z * n_obj.x + c1.z * n_obj.y) + c2.z * n_obj.z);   // :267
        nrm = normalize3(n_w);                                           // :268
        const uint32_t rslot = inst * (uint32_t)s.max_sub + geom;
        f3 surf = mk3(s.base_color[rslot]);  // :262-269
        if (CHAIN) chain_out = fp.bounce == 0 ? rslot : (chain_in & 0xFFFFu) | (rslot << 16);
        int idx = (int)(q2load(&io.seeds[pix]) + fp.sampleIndex);      // pix = sub * npix + pixel: the table entry already holds + sub
        const int dim0 = 2 + fp.bounce * 5;
        norg = P + nrm * 1e-3f;                                          // :350, :390
        color = mk3(C);
        bool diffuse = true;
        if (MATERIALS) {
            // the materials extension (renderer option materials = 1; not in raytracingKernel — README.md:8 lists it as open work; the
            // fields are ShaderTypes.h:99-107).  Semantics: DESIGN.md "Materials extension"; the CPU checker restates this block expression by expression.
            const float4 *__restrict__ mp = s.materials + 3 * (size_t)(inst * (uint32_t)s.max_sub + geom);
            const float4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
            const f3 em = color * mk3(m2);
            // bounce 0: pix == the sample index and the zero (k_shade) went through `sample_primary`; the emission follows through the SAME
            // pointer (both are __restrict__: two names for one address would leave the order of the two stores to the compiler)
            if (io.sample_primary) io.sample_primary[pix] = make_float4(0.0f + em.x, 0.0f + em.y, 0.0f + em.z, 0.0f);
            else { const float4 acc = io.sample[pix]; io.sample[pix] = make_float4(acc.x + em.x, acc.y + em.y, acc.z + em.z, 0.0f); }
            const float ul = halton_dev(idx, 2 + 5 * fp.max_bounces + fp.bounce);
            const f3 spec = mk3(m1);
            const float kd = fmaxf(surf.x, fmaxf(surf.y, surf.z)), ks = fmaxf(spec.x, fmaxf(spec.y, spec.z));
            const float dis = m0.w, ns = m1.w, ni = m2.w;
            const float trn = (dis > 0.0f && dis < 1.0f && ni > 0.0f) ? 1.0f - dis : 0.0f;
            if (ul < trn) {                                              // dielectric interface
                const float u2 = ul / trn;
                const f3 dir = mk3(B);
                const float cd = dot3(dir, nrm);
                const bool entering = cd < 0.0f;
                const f3 nn = entering ? nrm : neg3(nrm);
                const float eta = entering ? 1.0f / ni : ni;
                const float cosi = entering ? -cd : cd;
                const float sin2t = (eta * eta) * (1.0f - cosi * cosi);
                float r0 = (1.0f - ni) / (1.0f + ni); r0 = r0 * r0;
                float F = 1.0f, cost = 0.0f;
                if (sin2t < 1.0f) { cost = __builtin_sqrtf(1.0f - sin2t); const float c = entering ? cosi : cost; const float x = 1.0f - c; const float x2 = x * x; F = r0 + (1.0f - r0) * ((x2 * x2) * x); }
                f3 nd;
                if (u2 < F) { nd = dir + nn * (2.0f * cosi); norg = P + nn * 1e-3f; }
                else { nd = dir * eta + nn * (eta * cosi - cost); norg = P + nn * -1e-3f; }
                ndir = normalize3(nd);
                diffuse = false; special = true;
                want_next = fp.bounce + 1 < fp.max_bounces;
            } else {
                const float ud = trn > 0.0f ? (ul - trn) / (1.0f - trn) : ul;
                const float ps = (ks > 0.0f && ns > 0.0f) ? ks / (ks + kd) : 0.0f;
                if (ud < ps) {                                           // specular lobe
                    const float hx = halton_dev(idx, dim0 + 3), hy = halton_dev(idx, dim0 + 4);
                    const float a2 = 2.0f / (ns + 2.0f);
                    const float ct2 = (1.0f - hy) / (1.0f + (a2 - 1.0f) * hy);
                    const float ct = __builtin_sqrtf(ct2), st = __builtin_sqrtf(1.0f - ct2);
                    float sp_, cp_; sincos_2pi_dev(hx, sp_, cp_);
                    const f3 hw = align_hemisphere_dev(mk3(st * cp_, ct, st * sp_), nrm);
                    const f3 dir = mk3(B);
                    const float dh = dot3(dir, hw);
                    const f3 wi = dir - hw * (2.0f * dh);
                    diffuse = false; special = true;
                    if (dot3(wi, nrm) > 0.0f) {
                        color = color * (spec * (1.0f / ps));
                        ndir = normalize3(wi);
                        want_next = fp.bounce + 1 < fp.max_bounces;
                    }                                                    // else: sampled below the surface, the path is absorbed
                } else if (ps > 0.0f) surf = surf * (1.0f / (1.0f - ps));
            }
        }
        if (diffuse) {
            float ls = B0TAB ? halton_b0(fp, idx, 2) : halton_dev(idx, dim0 + 0);               // :272 (B0TAB: bounce 0 — dimensions 2 .. 6, from the table when there is one)
            int li = min((int)(ls * (float)fp.lightCount), fp.lightCount - 1);   // :273
            const LightDev L = s.lights[li];
            int ltype = __float_as_int(L.position.w);
            if (ltype == MRTLightTypeAreaLight) {                            // :281-290, :94-128
                float ax = (B0TAB ? halton_b0(fp, idx, 3) : halton_dev(idx, dim0 + 1)) * 2.0f - 1.0f;
                float ay = (B0TAB ? halton_b0(fp, idx, 4) : halton_dev(idx, dim0 + 2)) * 2.0f - 1.0f;
                f3 sp = (mk3(L.position) + mk3(L.right) * ax) + mk3(L.up) * ay;
                ldir = sp - P;
                ldist = length3(ldir);
                float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                ldir = ldir * inv;
                lcol = mk3(L.color) * (inv * inv);
                lcol = lcol * saturatef(dot3(neg3(ldir), mk3(L.forward)));
            } else if (ltype == MRTLightTypeSpotlight) {                     // :292-316
                ldir = mk3(L.position) - P;
                ldist = length3(ldir);
                float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                ldir = ldir * inv;
                lcol = mk3(0, 0, 0);
                float spot = dot3(neg3(ldir), mk3(L.dirn));
                if (spot > L.dirn.w) lcol = (mk3(L.color) * inv) * inv;
            } else if (ltype == MRTLightTypePointlight) {                    // :317-322
                ldir = mk3(L.position) - P;
                ldist = length3(ldir);
                float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                ldir = ldir * inv;
                lcol = (mk3(L.color) * inv) * inv;
            } else {                                                         // :323-327
                ldir = neg3(mk3(L.dirn));
                ldist = __builtin_inff();
                lcol = mk3(L.color);
            }
            lcol = lcol * saturatef(dot3(nrm, ldir));                        // :331
            lcol = lcol * (float)fp.lightCount;                              // :335
            color = mk3(C) * surf;                                           // :339
            want_shadow = length3(lcol) > 0.0001f;                           // :341
            want_next = fp.bounce + 1 < fp.max_bounces;
            if (want_next) {
                float hx = B0TAB ? halton_b0(fp, idx, 5) : halton_dev(idx, dim0 + 3), hy = B0TAB ? halton_b0(fp, idx, 6) : halton_dev(idx, dim0 + 4);   // :384-385
                ndir = align_hemisphere_dev(sample_cosine_hemisphere_dev(hx, hy), nrm);  // :387-388
            }
        }
    }
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // MATERIALS: the block's next rays are queued by lobe class — diffuse first, specular / refracted behind them — so that the waves of the
    // next traversal launch see rays of one kind (README.md:9 "sorting ... to reduce divergence"; the reference's stub is Raytracing.metal:178-197)
    const unsigned long long m_sh = __ballot(want_shadow), m_nx = __ballot(want_next && !(MATERIALS && special)), m_sp = MATERIALS ? __ballot(want_next && special) : 0ull;
    if (lane == 0) { sh.w_shadow[wv] = (uint32_t)__popcll(m_sh); sh.w_next[wv] = (uint32_t)__popcll(m_nx); if (MATERIALS) sh.w_spec[wv] = (uint32_t)__popcll(m_sp); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tn = 0, ts = 0, tp = 0;
        for (int k = 0; k < SHADE_WAVES; k++) { uint32_t a = sh.w_next[k], b = sh.w_shadow[k]; sh.w_next[k] = tn; sh.w_shadow[k] = ts; tn += a; ts += b; }
        if (MATERIALS) for (int k = 0; k < SHADE_WAVES; k++) { uint32_t a = sh.w_spec[k]; sh.w_spec[k] = tn + tp; tp += a; }
        sh.blk_base = (tn | ts | tp) ? atomicAdd(io.count_out, ((unsigned long long)ts << 32) | (tn + tp)) : 0ull;
    }
    __syncthreads();
    const unsigned long long base = sh.blk_base;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t ss = 0, ns = 0;
    if (want_shadow) {
        ss = (uint32_t)(base >> 32) + sh.w_shadow[wv] + (uint32_t)__popcll(m_sh & lt);
        f3 so = P + nrm * 1e-3f;                                         // :350
        f3 con = lcol * color;                                           // :372
        qstore(&io.srayA[ss], make_float4(so.x, so.y, so.z, ldist - 1e-3f));        // :356
        qstore(&io.srayB[ss], make_float4(ldir.x, ldir.y, ldir.z, __uint_as_float(pix)));
        if (PLANES) q2store(&io.scon[pix], make_float4(con.x, con.y, con.z, 0.0f)); else qstore(&io.scon[ss], make_float4(con.x, con.y, con.z, 0.0f));
    }
    if (want_next) {
        const bool sp = MATERIALS && special;
        ns = (uint32_t)base + (sp ? sh.w_spec[wv] + (uint32_t)__popcll(m_sp & lt) : sh.w_next[wv] + (uint32_t)__popcll(m_nx & lt));
        qstore(&io.nrayA[ns], make_float4(norg.x, norg.y, norg.z, CHAIN ? __uint_as_float(chain_out) : __builtin_inff()));      // :390 (tmax = inf either way, see FrameParams::chain)
        qstore(&io.nrayB[ns], make_float4(ndir.x, ndir.y, ndir.z, __uint_as_float(pix)));   // :391
        if (!CHAIN) qstore(&io.nthr[ns], make_float4(color.x, color.y, color.z, 0.0f));
    }
}

// thread -> (slot, sub-frame) of a bounce-0 launch.  Plain: grid = (blocks over one sub-frame's slots, sub-frames).  frame_bundle
// (wave-uniform): the lanes 8 b .. 8 b + 7 of a wave are eight sub-frames of ONE slot, the launch is one grid row.
MRT_DEV bool bounce0_slot(const FrameParams &fp, uint32_t &slot, uint32_t &sub) {
    sub = blockIdx.y; slot = blockIdx.x * SHADE_THREADS + threadIdx.x;
    if (!fp.frame_bundle) return true;
    const uint32_t lane = threadIdx.x & 63u, wave = slot >> 6;
    const uint32_t bi = (lane * fp.bundle_magic) >> 16, q = wave * fp.bundle_per_wave + bi;          // the lane's bundle: bi-th of its wave, q-th of the launch
    slot = q / fp.bundle_groups; sub = (q - slot * fp.bundle_groups) * fp.bundle_w + (lane - bi * fp.bundle_w);
    return bi < fp.bundle_per_wave && sub < (uint32_t)fp.batch;
}

// Bounce 0 of the fused pipeline (Renderer::fuse_primary): the primary ray is generated and traced HERE and its hit shaded from
// registers — no hit record, no direction record, one launch less per pass.  Shadow planes + throughput chain always.
//   WALK 1  the rope walk (scenes without the 8-wide layout), k_trace_primary<false>'s body
//   WALK 2  one ray per lane on the 8-wide layout (traverse_wide_lane), the wave's stack in dynamic LDS
//   WALK 3  two-level scene: both levels on one stack (traverse_wide_lane_two_level); the hint is (packet | instance << 24)
template <int WALK>
__global__ void __launch_bounds__(SHADE_THREADS, WALK >= 2 ? MRT_SHADE_WIDE_WAVES : MRT_SHADE_WAVES) k_shade_primary(SceneView s, FrameParams fp, SHADE_IO_PARAMS) {
    const ShadeIO io = SHADE_IO_GATHER;
    __shared__ ShadeShared sh;
    extern __shared__ uint32_t shade_stk[];          // WALK >= 2: SHADE_WAVES x (wide-tree depth x WIDE_STACK_LEVEL_BYTES) [+ SHADE_WAVES x 1 KB of hit words]
    uint32_t slot, sub;
    const bool in_batch = bounce0_slot(fp, slot, sub);
    const uint32_t spix = sub * io.capacity + slot;          // sample index (seed table, sample buffer, contribution planes)
    int px_x = 0, px_y = 0;
    const bool active = slot < io.capacity && in_batch && slot_to_pixel(fp, slot, px_x, px_y);
    float4 H = make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)), Bprim = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
    if (active) {
        f3 org, dir;
        primary_ray(fp, io.seeds, spix, px_x, px_y, org, dir);
        Bprim = make_float4(dir.x, dir.y, dir.z, __uint_as_float(spix));
        const uint32_t pixel = (uint32_t)px_y * (uint32_t)fp.width + (uint32_t)px_x;
        uint32_t *const stk = shade_stk + (threadIdx.x >> 6) * fp.wide_stack_words;
        TravHit h;
        bool hit;
        // the triangle this pixel hit in an earlier frame is tested first: the jittered ray most often hits it again, and the walk then starts with
        // the right distance bound instead of discovering it.  Any packet is a legal guess (a wrong one is one wasted test); the result is unchanged.
        if (WALK == 3) {
            if (io.hint != nullptr) {
                const uint32_t guess = io.hint[pixel];
                float t0 = __builtin_inff(); uint32_t seed = 0xFFFFFFFFu;
                const uint32_t gpk = guess & 0xFFFFFFu, gin = guess >> 24;
                if (guess != 0xFFFFFFFFu && gin < s.num_inst) {
                    const InstanceDev &I = s.inst[gin];
                    if (gpk - I.packet_base < I.ntri) {           // a legal guess names a packet of its instance's BLAS (a stale one — moved instances — is one wasted test or none)
                        const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)gpk;
                        float t, U, V, ad;
                        if (tri_test(pk[0], pk[1], pk[2], to_object_point(I, org), to_object_dir(I, dir), 0.0f, __builtin_inff(), t, U, V, ad)) { t0 = t; seed = guess; }
                    }
                }
                hit = traverse_wide_lane_two_level<true>(s, org, dir, t0, seed, h, stk);
                if (hit && h.pk != guess) io.hint[pixel] = h.pk;
            }
            else hit = traverse_wide_lane_two_level<false>(s, org, dir, __builtin_inff(), 0xFFFFFFFFu, h, stk);
        }
        else if (WALK == 2) {
            if (io.hint != nullptr) {
                const uint32_t guess = io.hint[pixel];
                float t0 = __builtin_inff(); uint32_t seed = 0xFFFFFFFFu;
                TravHit sh_; sh_.U = 0.0f; sh_.V = 0.0f; sh_.ad = 1.0f; sh_.gid = 0xFFFFFFFFu;
                if (guess < s.num_wpackets) {
                    const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)guess;
                    const float4 q0_ = pk[0];
                    float t, U, V, ad;
                    if (tri_test(q0_, pk[1], pk[2], org, dir, 0.0f, __builtin_inff(), t, U, V, ad)) { t0 = t; seed = guess; sh_.U = U; sh_.V = V; sh_.ad = ad; sh_.gid = __float_as_uint(q0_.w); }
                }
#if MRT_LANE_HIT_LDS
                hit = traverse_wide_lane<true>(s, org, dir, t0, seed, h, stk, reinterpret_cast<float *>(shade_stk + SHADE_WAVES * fp.wide_stack_words + (threadIdx.x >> 6) * 256u), &sh_);
#else
                hit = traverse_wide_lane<true>(s, org, dir, t0, seed, h, stk);
#endif
                if (h.pk != guess) io.hint[pixel] = h.pk;
            }
            else hit = traverse_wide_lane<false>(s, org, dir, __builtin_inff(), 0xFFFFFFFFu, h, stk);
        }
        else if (io.hint != nullptr) {
            const uint32_t guess = io.hint[pixel];
            h.t = __builtin_inff(); h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu; h.pk = 0xFFFFFFFFu;
            if (guess < s.num_tris) {
                const float4 *__restrict__ pk = s.packets + 3 * (size_t)guess;
                const float4 q0 = pk[0];
                float t, U, V, ad;
                if (tri_test(q0, pk[1], pk[2], org, dir, 0.0f, __builtin_inff(), t, U, V, ad)) { h.t = t; h.U = U; h.V = V; h.ad = ad; h.gid = __float_as_uint(q0.w); h.pk = guess; }
            }
            hit = traverse<false, false, false, true>(s, org, dir, 0.0f, h.t, h);
            if (h.pk != guess) io.hint[pixel] = h.pk;
        }
        else hit = traverse<false>(s, org, dir, 0.0f, __builtin_inff(), h);
        if (hit) H = make_float4(h.t, h.U / h.ad, h.V / h.ad, __uint_as_float(h.gid));
    }
    shade_entry<false, true, true, true>(s, fp, io, sh, spix, H, active, Bprim);
}

// Bounces >= 1 (renderer option shade_pack): half the entries of a bounce queue are rays that missed — their lanes would sit out
// the whole kernel (41 % of the lanes per VALU instruction, profiles/r05_summary.json).  A workgroup takes fp.pack_range
// consecutive entries, compacts the HITS into a ring in LDS 256 entries at a time and shades 256 of them per round, every lane busy.
template <bool MATERIALS, bool CHAIN, bool PLANES, bool PAIRS>
__global__ void __launch_bounds__(SHADE_THREADS, MRT_SHADE_WAVES) k_shade_pack(SceneView s, FrameParams fp, SHADE_IO_PARAMS) {
    const ShadeIO io = SHADE_IO_GATHER;
    __shared__ ShadeShared sh;
    __shared__ uint32_t p_idx[2 * SHADE_THREADS], p_w[SHADE_WAVES];
    __shared__ float4 p_hit[2 * SHADE_THREADS];
    const uint32_t n = io.count_in ? (uint32_t)*io.count_in : io.capacity;
    const float4 miss = make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)), noB = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
    const uint32_t begin = blockIdx.x * fp.pack_range, end = min(n, begin + fp.pack_range);
    const uint32_t lane_ = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
    uint32_t head = 0, tail = 0, cur = begin;                  // the ring's positions only ever grow; entry k lives in slot k & (2 * SHADE_THREADS - 1); all three are workgroup-uniform
    for (;;) {
        const uint32_t pending = tail - head;
        if (pending < (uint32_t)SHADE_THREADS && cur < end) {          // room for 256 more: the next 256 entries' hits join the ring
            const uint32_t e = cur + threadIdx.x;
            float4 He = miss;
            if (PAIRS) {          // the entry's key: kept as two words of the record, turned into a hit record in the round that shades it
                const unsigned long long key = e < end ? __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(io.hits) + e) : ~0ull;
                He = make_float4(__uint_as_float((uint32_t)key), __uint_as_float((uint32_t)(key >> 32)), 0.0f, __uint_as_float(key != ~0ull ? 0u : 0xFFFFFFFFu));
            }
            else if (e < end) He = qload(&io.hits[e]);
            const bool a = __float_as_uint(He.w) != 0xFFFFFFFFu;
            const unsigned long long m = __ballot(a);
            if (lane_ == 0) p_w[wv_] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t off = 0, tot = 0;
            for (int k = 0; k < SHADE_WAVES; k++) { const uint32_t c = p_w[k]; if ((uint32_t)k < wv_) off += c; tot += c; }
            if (a) { const uint32_t pos = (tail + off + (uint32_t)__popcll(m & ((1ull << lane_) - 1ull))) & (2u * SHADE_THREADS - 1u); p_idx[pos] = e; p_hit[pos] = He; }
            __syncthreads();
            tail += tot; cur += SHADE_THREADS;
            continue;
        }
        if (pending == 0u) break;
        const uint32_t take = min(pending, (uint32_t)SHADE_THREADS);
        const bool act = threadIdx.x < take;
        const uint32_t pos = (head + threadIdx.x) & (2u * SHADE_THREADS - 1u);
        const uint32_t ie = act ? p_idx[pos] : 0u;
        float4 He = act ? p_hit[pos] : miss;
        if (PAIRS && act) He = pairs_hit(s, io, ie, (unsigned long long)__float_as_uint(He.x) | ((unsigned long long)__float_as_uint(He.y) << 32));
        head += take;
        shade_entry<MATERIALS, CHAIN, PLANES, false>(s, fp, io, sh, ie, He, act, noB);
    }
}

// One queue entry per thread, hit or miss: bounce 0 from the hit records of a primary launch (grid = (blocks over one sub-frame's
// slots, sub-frames)), any bounce of the general pipeline (materials, > 3 bounces, no 8-wide layout), and the A/B of the packed form.
template <bool MATERIALS, bool CHAIN, bool PLANES, bool PAIRS>
__global__ void __launch_bounds__(SHADE_THREADS, MRT_SHADE_WAVES) k_shade(SceneView s, FrameParams fp, SHADE_IO_PARAMS) {
    const ShadeIO io = SHADE_IO_GATHER;
    __shared__ ShadeShared sh;
    const uint32_t sub = io.sample_primary ? blockIdx.y : 0u, slot = blockIdx.x * SHADE_THREADS + threadIdx.x;
    const uint32_t i = sub * io.capacity + slot;
    const uint32_t n = io.count_in ? (uint32_t)*io.count_in : io.capacity;
    bool active = slot < n;
    if (io.sample_primary) {
        int px_x, px_y;
        active = active && slot_to_pixel(fp, slot, px_x, px_y);
        if (active && !PLANES) q2store(&io.sample_primary[i], make_float4(0.0f, 0.0f, 0.0f, 0.0f));   // Raytracing.metal:227 (sample index = sub * capacity + slot)
    }
    float4 H = make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
    if (active && PAIRS) H = pairs_hit(s, io, i, __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(io.hits) + i));
    else if (active) H = qload(&io.hits[i]);
    shade_entry<MATERIALS, CHAIN, PLANES, false>(s, fp, io, sh, i, H, active, make_float4(0.0f, 0.0f, 1.0f, 0.0f));
}

}  // namespace
}  // namespace mrt
