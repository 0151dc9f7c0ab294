// megakernel.h — one launch per frame: whole paths per lane of persistent waves (renderer option megakernel = 1).
// Included by renderer.hip inside namespace mrt { namespace { ... } } after k_shade's helpers (primary_ray, slot_to_pixel) and traverse_wide.h.
#pragma once

// The reference is ONE kernel per frame (Raytracing.metal:156-405).  The wavefront pipeline above is faster in throughput (its shading runs at
// full lane width, its traversal launches are large) but a single frame pays eight dependent launches, each with its own ramp-up and tail:
// 2.07 ms for a frame whose arithmetic is 0.65 ms.  Here every lane of a persistent wave carries a whole PATH: primary ray -> closest hit ->
// shade -> shadow ray (any hit) -> bounce ray -> ... on the 8-wide layout with the LDS stack of traverse_wide.h; finished rays are serviced
// (shaded, turned into their follow-up ray, or replaced by the next pixel) whenever a quarter of the wave is waiting, and the pixel's running
// average is written when its path ends (Raytracing.metal:394-403).  No ray queues, no k_shade, no k_accumulate.
// One pixel belongs to one lane from start to end, so no atomics on the image; pixels are pulled 64 slots at a time from a shared counter.
// Restates the shading of k_shade<false> (same expressions, same order: the image is bit-identical to the wavefront pipeline's and the oracle's).
__global__ void __launch_bounds__(64, 4) k_megakernel(SceneView s, FrameParams fp, const uint32_t *__restrict__ seeds, const float4 *__restrict__ prev, float4 *__restrict__ dst,
                                                      uint32_t *__restrict__ work, unsigned long long *__restrict__ totals, uint32_t primary) {
    extern __shared__ uint32_t stk_dyn[];
    uint32_t *stack = stk_dyn;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&totals[2], (unsigned long long)primary);
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t n_slots = fp.capacity;
    // prefetched pixels: slot batch_base + lane -> primary direction + pixel index (0xFFFFFFFF: slot outside the image)
    f3 pd = mk3(0, 0, 1); uint32_t ppix = 0xFFFFFFFFu;
    uint32_t batch_n = 0, batch_used = 0, batch_base = 0;
    bool more = true;
    // path state of the lane
    bool has_path = false, live = false, is_shadow = false, pending = false;
    uint32_t pix = 0, pslot = 0; int bounce = 0;      // pixel (accumulation targets) and slot (seed table) of the lane's path
    f3 thr = mk3(1, 1, 1), rad = mk3(0, 0, 0), con = mk3(0, 0, 0), ndir = mk3(0, 1, 0);
    uint32_t n_closest = 0, n_shadow = 0;
    // ray + traversal state (traverse_wide_stream)
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1); float ix = 0, iy = 0, iz = 0; bool nx = false, ny = false, nz = false; uint32_t oct = 0;
    float best_t = 0.0f; uint32_t best_pk = 0xFFFFFFFFu;
    uint32_t g_base = 0, g_mask = 0, t_base = 0, t_mask = 0;
    auto start_ray = [&](f3 ro, f3 rd, float tmax) {
        o = ro; d = rd; ix = box_inv(rd.x); iy = box_inv(rd.y); iz = box_inv(rd.z);
        nx = rd.x < 0.0f; ny = rd.y < 0.0f; nz = rd.z < 0.0f; oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
        best_t = tmax; best_pk = 0xFFFFFFFFu;
        g_base = 0; g_mask = s.num_wnodes != 0 ? 0x100u : 0u; t_base = 0; t_mask = 0;
        live = true;
    };
    for (;;) {
        const unsigned long long m_wait = __ballot(!live);
        if ((uint32_t)__popcll(m_wait) >= (uint32_t)WIDE_REFILL_AT || m_wait == ~0ull) {
            // ---- 1. lanes whose ray has finished
            if (has_path && !live) {
                bool end_path = false;
                if (is_shadow) {
                    if (best_pk == 0xFFFFFFFFu) rad = rad + con;                                   // :371-373 (unoccluded)
                    is_shadow = false;
                    if (pending) { pending = false; bounce++; n_closest++; start_ray(o, ndir, __builtin_inff()); }      // the bounce ray leaves from the same offset point (:390)
                    else end_path = true;
                } else if (best_pk == 0xFFFFFFFFu) end_path = true;                                // :246-247 miss terminates the path
                else {
                    // hit record of the winning triangle (recomputed: same arithmetic as the traversal's test)
                    const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)best_pk;
                    const float4 q0 = pk[0];
                    float t_, U, V, ad;
                    (void)tri_test(q0, pk[1], pk[2], o, d, 0.0f, __builtin_inff(), t_, U, V, ad);
                    const uint32_t gid = __float_as_uint(q0.w);
                    const float bu = U / ad, bv = V / ad;
                    const uint4 ts = s.tri_shade[gid];
                    const uint32_t inst = ts.w >> 16, geom = ts.w & 0xFFFFu;
                    const f3 P = o + d * best_t;                                                   // :261
                    const float bw = 1.0f - bu - bv;                                               // :63-64
                    const f3 n_obj = (bu * mk3(s.normals[ts.y]) + bv * mk3(s.normals[ts.z])) + bw * mk3(s.normals[ts.x]);   // :66-72
                    const f3 c0 = mk3(s.inst_cols[inst * 4 + 0]), c1 = mk3(s.inst_cols[inst * 4 + 1]), c2 = mk3(s.inst_cols[inst * 4 + 2]);
                    const f3 n_w = mk3((c0.x * n_obj.x + c1.x * n_obj.y) + c2.x * n_obj.z,
                                       (c0.y * n_obj.x + c1.y * n_obj.y) + c2.y * n_obj.z,
                                       (c0.z * n_obj.x + c1.z * n_obj.y) + c2.z * n_obj.z);        // :267
                    const f3 nrm = normalize3(n_w);                                                // :268
                    const f3 surf = mk3(s.base_color[inst * (uint32_t)s.max_sub + geom]);          // :262-269
                    const int idx = (int)(seeds[pslot] + fp.sampleIndex);
                    const int dim0 = 2 + bounce * 5;
                    const float ls = halton_dev(idx, dim0 + 0);                                    // :272
                    const int li = min((int)(ls * (float)fp.lightCount), fp.lightCount - 1);       // :273
                    const LightDev L = s.lights[li];
                    const int ltype = __float_as_int(L.position.w);
                    f3 ldir, lcol; float ldist;
                    if (ltype == MRTLightTypeAreaLight) {                                          // :281-290, :94-128
                        const float ax = halton_dev(idx, dim0 + 1) * 2.0f - 1.0f;
                        const float ay = halton_dev(idx, dim0 + 2) * 2.0f - 1.0f;
                        const f3 sp = (mk3(L.position) + mk3(L.right) * ax) + mk3(L.up) * ay;
                        ldir = sp - P;
                        ldist = length3(ldir);
                        const float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                        ldir = ldir * inv;
                        lcol = mk3(L.color) * (inv * inv);
                        lcol = lcol * saturatef(dot3(neg3(ldir), mk3(L.forward)));
                    } else if (ltype == MRTLightTypeSpotlight) {                                   // :292-316
                        ldir = mk3(L.position) - P;
                        ldist = length3(ldir);
                        const float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                        ldir = ldir * inv;
                        lcol = mk3(0, 0, 0);
                        const float spot = dot3(neg3(ldir), mk3(L.dirn));
                        if (spot > L.dirn.w) lcol = (mk3(L.color) * inv) * inv;
                    } else if (ltype == MRTLightTypePointlight) {                                  // :317-322
                        ldir = mk3(L.position) - P;
                        ldist = length3(ldir);
                        const float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
                        ldir = ldir * inv;
                        lcol = (mk3(L.color) * inv) * inv;
                    } else {                                                                       // :323-327
                        ldir = neg3(mk3(L.dirn));
                        ldist = __builtin_inff();
                        lcol = mk3(L.color);
                    }
                    lcol = lcol * saturatef(dot3(nrm, ldir));                                      // :331
                    lcol = lcol * (float)fp.lightCount;                                            // :335
                    thr = thr * surf;                                                              // :339
                    const bool want_shadow = length3(lcol) > 0.0001f;                              // :341
                    const bool want_next = bounce + 1 < fp.max_bounces;
                    if (want_next) {
                        const float hx = halton_dev(idx, dim0 + 3), hy = halton_dev(idx, dim0 + 4);   // :384-385
                        ndir = align_hemisphere_dev(sample_cosine_hemisphere_dev(hx, hy), nrm);       // :387-388
                    }
                    const f3 off = P + nrm * 1e-3f;                                                // :350, :390
                    if (want_shadow) { con = lcol * thr; pending = want_next; is_shadow = true; n_shadow++; start_ray(off, ldir, ldist - 1e-3f); }   // :356, :372
                    else if (want_next) { bounce++; n_closest++; start_ray(off, ndir, __builtin_inff()); }
                    else end_path = true;
                }
                if (end_path) {                                                                    // :394-403
                    float4 c = make_float4(rad.x, rad.y, rad.z, 1.0f);
                    if (fp.frameIndex > 0) {
                        const float4 p = prev[pix];
                        const float fi = (float)fp.frameIndex, den = (float)(fp.frameIndex + 1);
                        c.x = (rad.x + p.x * fi) / den; c.y = (rad.y + p.y * fi) / den; c.z = (rad.z + p.z * fi) / den;
                    }
                    dst[pix] = c;
                    has_path = false;
                }
            }
            // ---- 2. lanes without a path take the next pixels
            if (batch_used >= batch_n && more) {                       // prefetch 64 pixel slots: primary rays generated by the whole wave
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work, 64u);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= n_slots) { more = false; batch_n = batch_used = 0; }
                else {
                    batch_n = min(64u, n_slots - base); batch_used = 0; batch_base = base;
                    int x, y; ppix = 0xFFFFFFFFu;
                    if (lane < batch_n && slot_to_pixel(fp, base + lane, x, y)) {
                        f3 org; primary_ray(fp, seeds, base + lane, x, y, org, pd);
                        ppix = (uint32_t)y * (uint32_t)fp.width + (uint32_t)x;
                    }
                }
            }
            const unsigned long long m_free = __ballot(!has_path);
            const uint32_t avail = batch_n - batch_used, n_free = (uint32_t)__popcll(m_free);
            if (avail != 0 && n_free != 0) {
                const uint32_t rank = (uint32_t)__popcll(m_free & lt);
                const bool take = !has_path && rank < avail;
                const int sl = (int)(take ? batch_used + rank : lane);
                const float dx_ = __shfl(pd.x, sl), dy_ = __shfl(pd.y, sl), dz_ = __shfl(pd.z, sl);
                const uint32_t px_ = (uint32_t)__shfl((int)ppix, sl);
                if (take && px_ != 0xFFFFFFFFu) {
                    has_path = true; is_shadow = false; pending = false; pix = px_; pslot = batch_base + (uint32_t)sl; bounce = 0;
                    thr = mk3(1.0f, 1.0f, 1.0f); rad = mk3(0.0f, 0.0f, 0.0f);                     // :226-227
                    n_closest++;
                    start_ray(mk3(fp.cam_pos), mk3(dx_, dy_, dz_), __builtin_inff());              // :214-221
                }
                batch_used += min(avail, n_free);
                continue;
            }
            if (__ballot(has_path) == 0ull) { if (!more) break; else continue; }
            if (__ballot(live) == 0ull) continue;      // everybody was serviced into a finished state again (cannot happen: a serviced lane is live or pathless)
        }
        // ---- traversal step: traverse_wide_stream's iteration (one memory round trip: node and triangle fetched together)
        const bool has_tri = live && t_mask != 0;
        const uint32_t t_rest = t_mask & (t_mask - 1u);
        bool want_node = live && t_rest == 0u;
        uint32_t pending_node = 0, tri_pk = 0;
        if (want_node) {
            if ((g_mask & 0xFF00u) == 0) {
                const uint32_t sp = g_mask >> 16;
                if (sp == 0) { want_node = false; if (!has_tri) live = false; }
                else { wstack_pop(stack, sp - 1u, lane, g_base, g_mask); g_mask |= (sp - 1u) << 16; }
            }
            if (want_node) {
                const uint32_t hits = (g_mask >> 8) & 0xFFu;
                const uint32_t b = (uint32_t)__ffs((int)hits) - 1u;
                g_mask &= ~(0x100u << b);
                const uint32_t slot = b ^ oct;
                pending_node = g_base + (uint32_t)__popc(g_mask & 0xFFu & ((1u << slot) - 1u));
            }
        }
        float4 r0, r1, r2, n0, n1, n2, n3, n4;
        asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r2.x), "=v"(r2.y), "=v"(r2.z));
        asm volatile("" : "=v"(n0.x), "=v"(n0.y), "=v"(n0.z), "=v"(n0.w), "=v"(n1.x), "=v"(n1.y), "=v"(n1.z), "=v"(n1.w), "=v"(n2.x), "=v"(n2.y), "=v"(n2.z), "=v"(n2.w));
        asm volatile("" : "=v"(n3.x), "=v"(n3.y), "=v"(n3.z), "=v"(n3.w), "=v"(n4.x), "=v"(n4.y), "=v"(n4.z), "=v"(n4.w));
        r1.w = 0.0f; r2.w = 0.0f;
        if (has_tri) {
            tri_pk = t_base + (uint32_t)__ffs((int)t_mask) - 1u;
            t_mask = t_rest;
            const float4 *__restrict__ pk = s.wpackets + WPK * (size_t)tri_pk;
            r0 = pk[0]; r1 = pk[1]; r2 = pk[2];
        }
        if (want_node) {
            const float4 *__restrict__ nd = s.wnodes + WNODE_STRIDE * (size_t)pending_node;
            n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3]; n4 = nd[4];
        }
        if (has_tri) {
            float t, U, V, ad;
            if (tri_test(r0, r1, r2, o, d, 0.0f, best_t, t, U, V, ad)) {
                if (is_shadow) { best_pk = tri_pk; live = false; }                                 // any hit: done
                else {
                    bool better = t < best_t || best_pk == 0xFFFFFFFFu;
                    if (!better) better = __float_as_uint(r0.w) < __float_as_uint(s.wpackets[WPK * (size_t)best_pk].w);   // ties go to the lowest id
                    if (better) { best_t = t; best_pk = tri_pk; }
                }
            }
        }
        if (want_node && live) {
            uint32_t node_hits, tri_hits;
            wide_node_test(n0, n1, n2, n3, n4, o, ix, iy, iz, nx, ny, nz, oct, 0.0f, best_t, node_hits, tri_hits);      // (the scaled form needs one more register: 128 -> spills)
            uint32_t sp = g_mask >> 16;
            if ((g_mask & 0xFF00u) != 0) { wstack_push(stack, sp, lane, g_base, g_mask & 0xFFFFu); sp++; }
            g_base = __float_as_uint(n1.x); g_mask = (sp << 16) | (node_hits << 8) | (__float_as_uint(n0.w) >> 24);
            t_base = __float_as_uint(n1.y); t_mask = tri_hits;
        }
    }
    // ray counters of the frame (Renderer::stats): one atomic per wave
    for (int ofs = 32; ofs > 0; ofs >>= 1) { n_closest += (uint32_t)__shfl_xor((int)n_closest, ofs); n_shadow += (uint32_t)__shfl_xor((int)n_shadow, ofs); }
    if (lane == 0) { atomicAdd(&totals[0], (unsigned long long)n_closest); atomicAdd(&totals[1], (unsigned long long)n_shadow); }
}

