// bvh_build.hip — on-device acceleration-structure build for gfx950.
//
// Replaces createAccelerationStructures + buildCompactedAccelerationStructures
// (Renderer.swift:184-214, Utilities.swift:29-85), for which the reference holds no code at all
// (Apple's MTLAccelerationStructure is an opaque driver object).  Instances are flattened into one
// world-space triangle soup — the reference never shares a primitive AS between instances
// (Renderer.swift:193-195), so this is semantically identical.
//
// Pipeline (all kernels on the caller's stream, no host round trips until the final stats read):
//   flatten   object-space vertices -> world-space Möller–Trumbore triangles, padded AABBs, centroid bounds
//   morton    63-bit Morton code of the AABB centre (21 bits per axis over the centroid bounds)
//   sort      hand-written LSD radix sort, 8-bit digits, 64-bit keys + 32-bit payload, stable
//   topology  builder 0: Karras 2012 binary radix tree (plain LBVH)
//             builder 1: PLOC (Meister & Bittner 2018): surface-area nearest-neighbour agglomeration in
//                        a window over the Morton order — the SAH-driven refinement of the LBVH order
//   refit     bottom-up AABBs, SAH cost, SAH leaf collapse (<= max_leaf triangles), near-child masks
//   emit      preorder node numbering, 8-octant escape ("rope") links, leaf-contiguous triangle packets
#include "scene_device.h"
#include "device_math.h"
#include <algorithm>
#include <cstring>
#include <cmath>
#include <chrono>
#include <thread>
#include <atomic>
#include <functional>

namespace mrt {
namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

struct SubRec { uint32_t tri_begin, tri_count, index_offset, vbase, inst, geom; };

__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float ord2f(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// ------------------------------------------------------------------ flatten
__global__ void k_flatten(const SubRec *__restrict__ recs, int nrec, const float *__restrict__ pos,
                          const uint32_t *__restrict__ indices, const float4 *__restrict__ inst_cols, uint32_t T,
                          float4 *__restrict__ tri_world, uint4 *__restrict__ tri_shade,
                          float4 *__restrict__ leaf_lo, float4 *__restrict__ leaf_hi, uint32_t *__restrict__ cbounds) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = {0, 0, 0};
    bool valid = gid < T;
    if (valid) {
        int lo = 0, hi = nrec - 1;                       // last record with tri_begin <= gid
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (recs[mid].tri_begin <= gid) lo = mid; else hi = mid - 1; }
        SubRec r = recs[lo];
        uint32_t p = gid - r.tri_begin;
        const uint32_t *ix = indices + r.index_offset + 3 * (size_t)p;
        uint32_t i0 = ix[0] + r.vbase, i1 = ix[1] + r.vbase, i2 = ix[2] + r.vbase;
        float4 c0 = inst_cols[r.inst * 4 + 0], c1 = inst_cols[r.inst * 4 + 1], c2 = inst_cols[r.inst * 4 + 2], c3 = inst_cols[r.inst * 4 + 3];
        f3 w[3];
        uint32_t vi[3] = {i0, i1, i2};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float x = pos[3 * (size_t)vi[k]], y = pos[3 * (size_t)vi[k] + 1], z = pos[3 * (size_t)vi[k] + 2];
            // world = M * (p,1), fused form of mrt-math v1
            w[k].x = __builtin_fmaf(c2.x, z, __builtin_fmaf(c1.x, y, c0.x * x)) + c3.x;
            w[k].y = __builtin_fmaf(c2.y, z, __builtin_fmaf(c1.y, y, c0.y * x)) + c3.y;
            w[k].z = __builtin_fmaf(c2.z, z, __builtin_fmaf(c1.z, y, c0.z * x)) + c3.z;
        }
        f3 e1 = w[1] - w[0], e2 = w[2] - w[0];
        tri_world[3 * (size_t)gid + 0] = make_float4(w[0].x, w[0].y, w[0].z, __uint_as_float(gid));
        tri_world[3 * (size_t)gid + 1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
        tri_world[3 * (size_t)gid + 2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
        tri_shade[gid] = make_uint4(i0, i1, i2, (r.inst << 16) | r.geom);
        float blo[3], bhi[3];
        blo[0] = fminf(w[0].x, fminf(w[1].x, w[2].x)); bhi[0] = fmaxf(w[0].x, fmaxf(w[1].x, w[2].x));
        blo[1] = fminf(w[0].y, fminf(w[1].y, w[2].y)); bhi[1] = fmaxf(w[0].y, fmaxf(w[1].y, w[2].y));
        blo[2] = fminf(w[0].z, fminf(w[1].z, w[2].z)); bhi[2] = fmaxf(w[0].z, fmaxf(w[1].z, w[2].z));
#pragma unroll
        for (int k = 0; k < 3; k++) {   // pad: the slab test must never reject what the triangle test accepts
            float m = fmaxf(fabsf(blo[k]), fabsf(bhi[k]));
            float e = 1e-5f * m + 1e-6f;
            blo[k] -= e; bhi[k] += e;
            c[k] = 0.5f * (blo[k] + bhi[k]);
        }
        leaf_lo[gid] = make_float4(blo[0], blo[1], blo[2], 0.0f);
        leaf_hi[gid] = make_float4(bhi[0], bhi[1], bhi[2], 0.0f);
    }
    // bounds of the centres: wave, then workgroup (LDS), then one set of atomics per workgroup — the six words take ~90 atomics per microsecond, and one set per
    // wave (14 K waves for 885 K triangles) was the whole duration of this kernel (0.95 ms)
    const float BIG = 3.0e38f;
    __shared__ float smn[3][16], smx[3][16];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float mn = wave_min(valid ? c[k] : BIG), mx = wave_max(valid ? c[k] : -BIG);
        if (lane == 0) { smn[k][wv] = mn; smx[k][wv] = mx; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = (int)threadIdx.x;
        float mn = BIG, mx = -BIG;
        for (uint32_t i = 0; i < nw; i++) { mn = fminf(mn, smn[k][i]); mx = fmaxf(mx, smx[k][i]); }
        if (mn <= mx) { atomicMin(&cbounds[k], f2ord(mn)); atomicMax(&cbounds[3 + k], f2ord(mx)); }
    }
}

// ------------------------------------------------------------------ triangle pre-splitting (scene option presplit)
// A triangle much longer than its neighbours (a sliver of a scanned mesh, a fin) has a box that overlaps hundreds of others: every ray through
// that box pays a triangle test, and the agglomerative builder cannot separate what the boxes do not separate.  Early split clipping (Ernst &
// Greiner 2007; Karras & Aila 2013 §4): such a triangle enters the build as k REFERENCES, one per slab of its box along its longest axis, each
// with the bounds of the part of the triangle inside that slab.  A reference is a leaf like any other; its packet is the whole triangle, so a
// ray may test a split triangle more than once and gets the same answer each time — the closest hit stays the minimum over (t, id), the image
// does not change.  k = ceil(extent / (presplit x mean extent of all triangles)), at most 32; a uniformly tessellated mesh is not split at all.
MRT_DEV float comp3(const f3 &v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }
// deterministic mean: sum of (largest box extent / scene extent) in 2^-30 fixed point
__global__ void k_extent_sum(const float4 *__restrict__ leaf_lo, const float4 *__restrict__ leaf_hi, const uint32_t *__restrict__ cbounds, uint32_t T, unsigned long long *__restrict__ sum) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (gid < T) {
        const float4 lo = leaf_lo[gid], hi = leaf_hi[gid];
        const float scene = fmaxf(fmaxf(ord2f(cbounds[3]) - ord2f(cbounds[0]), ord2f(cbounds[4]) - ord2f(cbounds[1])), ord2f(cbounds[5]) - ord2f(cbounds[2]));
        const float e = fmaxf(fmaxf(hi.x - lo.x, hi.y - lo.y), hi.z - lo.z);
        const float r = scene > 0.0f ? fminf(e / scene, 4.0f) : 0.0f;
        v = (unsigned long long)(r * 1073741824.0f);
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __shared__ unsigned long long sv[16];                        // one atomic per workgroup (see k_flatten)
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (uint32_t i = 0; i < ((blockDim.x + 63) >> 6); i++) tot += sv[i];
        if (tot) atomicAdd(sum, tot);
    }
}
MRT_DEV uint32_t split_pieces(float4 lo, float4 hi, float scene, unsigned long long sum, uint32_t T, float factor, uint32_t max_pieces, int &ax, float &ext) {
    const float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
    ax = 0; ext = ex; if (ey > ext) { ext = ey; ax = 1; } if (ez > ext) { ext = ez; ax = 2; }
    const float mean = (float)((double)sum / 1073741824.0 / (double)T) * scene;
    if (!(mean > 0.0f) || !(ext > factor * mean)) return 1u;
    const float k = ceilf(ext / (factor * mean));
    return (uint32_t)fminf(fmaxf(k, 1.0f), (float)max_pieces);
}
__global__ void k_split_count(const float4 *__restrict__ tri_world, const float4 *__restrict__ leaf_lo, const float4 *__restrict__ leaf_hi, const uint32_t *__restrict__ cbounds, uint32_t T,
                              const unsigned long long *__restrict__ sum, float factor, uint32_t max_pieces, uint32_t *__restrict__ count) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= T) return;
    const float scene = fmaxf(fmaxf(ord2f(cbounds[3]) - ord2f(cbounds[0]), ord2f(cbounds[4]) - ord2f(cbounds[1])), ord2f(cbounds[5]) - ord2f(cbounds[2]));
    int ax; float ext;
    uint32_t k = split_pieces(leaf_lo[gid], leaf_hi[gid], scene, *sum, T, factor, max_pieces, ax, ext);
    if (k > 1u) {
        // only SLIVERS are split: longest edge more than eight times the height over it.  Slabs across the long axis of a sliver are short pieces of it; slabs
        // across a large but fat triangle (a floor quad) would be strips as long as the triangle — boxes worse than the one they replace (measured: DragonScene
        // 11.3 -> 8.8 Grays/s when its walls and floor were cut into 32 strips each)
        const f3 e1 = mk3(tri_world[3 * (size_t)gid + 1]), e2 = mk3(tri_world[3 * (size_t)gid + 2]), e3 = e2 - e1;
        const f3 c = cross3(e1, e2);
        const float area2 = __builtin_sqrtf(dot3(c, c));
        const float l2 = fmaxf(fmaxf(dot3(e1, e1), dot3(e2, e2)), dot3(e3, e3));
        if (!(l2 > 8.0f * area2)) k = 1u;
    }
    count[gid] = k;
}
__global__ void k_split_emit(const float4 *__restrict__ tri_world, const float4 *__restrict__ leaf_lo, const float4 *__restrict__ leaf_hi, const uint32_t *__restrict__ count,
                             const uint32_t *__restrict__ offset, uint32_t T, float4 *__restrict__ ref_lo, float4 *__restrict__ ref_hi, uint32_t *__restrict__ ref_tri) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= T) return;
    const uint32_t k = count[gid], base = offset[gid];
    const float4 blo = leaf_lo[gid], bhi = leaf_hi[gid];
    if (k <= 1u) { ref_lo[base] = blo; ref_hi[base] = bhi; ref_tri[base] = gid; return; }
    const float4 p0 = tri_world[3 * (size_t)gid], p1 = tri_world[3 * (size_t)gid + 1], p2 = tri_world[3 * (size_t)gid + 2];
    const f3 v[3] = {mk3(p0), mk3(p0) + mk3(p1), mk3(p0) + mk3(p2)};       // v0 + e is within an ulp of the vertex: far inside the padding below
    const float ex = bhi.x - blo.x, ey = bhi.y - blo.y, ez = bhi.z - blo.z;
    int ax = 0; float ext = ex; if (ey > ext) { ext = ey; ax = 1; } if (ez > ext) { ext = ez; ax = 2; }
    const float lo_ax = ax == 0 ? blo.x : ax == 1 ? blo.y : blo.z, hi_ax = ax == 0 ? bhi.x : ax == 1 ? bhi.y : bhi.z;
    for (uint32_t p = 0; p < k; p++) {
        // the same expression gives piece p's upper and piece p + 1's lower bound: no gap between neighbours
        const float a = p == 0 ? lo_ax : lo_ax + ext * ((float)p / (float)k), b = p + 1 == k ? hi_ax : lo_ax + ext * ((float)(p + 1) / (float)k);
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        auto take = [&](f3 q) { lo[0] = fminf(lo[0], q.x); lo[1] = fminf(lo[1], q.y); lo[2] = fminf(lo[2], q.z); hi[0] = fmaxf(hi[0], q.x); hi[1] = fmaxf(hi[1], q.y); hi[2] = fmaxf(hi[2], q.z); };
        for (int i = 0; i < 3; i++) {
            const f3 vi = v[i], vj = v[(i + 1) % 3];
            const float ci = comp3(vi, ax), cj = comp3(vj, ax);
            if (ci >= a && ci <= b) take(vi);
            for (int s_ = 0; s_ < 2; s_++) {
                const float pl = s_ ? b : a;
                if ((ci < pl) != (cj < pl)) {
                    const float t = (pl - ci) / (cj - ci);
                    f3 q = vi + (vj - vi) * t;
                    if (ax == 0) q.x = pl; else if (ax == 1) q.y = pl; else q.z = pl;
                    take(q);
                }
            }
        }
        if (lo[0] > hi[0]) { lo[0] = blo.x; lo[1] = blo.y; lo[2] = blo.z; hi[0] = bhi.x; hi[1] = bhi.y; hi[2] = bhi.z; lo[ax] = a; hi[ax] = b; }      // (rounding left the slab empty: its part of the box)
        for (int c = 0; c < 3; c++) {       // pad as k_flatten does, and never beyond the triangle's own padded box
            const float m = fmaxf(fabsf(lo[c]), fabsf(hi[c])), e = 1e-5f * m + 1e-6f;
            lo[c] -= e; hi[c] += e;
        }
        ref_lo[base + p] = make_float4(fmaxf(lo[0], blo.x), fmaxf(lo[1], blo.y), fmaxf(lo[2], blo.z), 0.0f);
        ref_hi[base + p] = make_float4(fminf(hi[0], bhi.x), fminf(hi[1], bhi.y), fminf(hi[2], bhi.z), 0.0f);
        ref_tri[base + p] = gid;
    }
}

// ------------------------------------------------------------------ morton
__device__ __forceinline__ uint64_t spread21(uint64_t x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ void k_morton(const float4 *__restrict__ leaf_lo, const float4 *__restrict__ leaf_hi, const uint32_t *__restrict__ cbounds,
                         uint32_t T, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= T) return;
    float4 lo = leaf_lo[gid], hi = leaf_hi[gid];
    float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    uint64_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float mn = ord2f(cbounds[k]), mx = ord2f(cbounds[3 + k]);
        float ext = mx - mn;
        float n = ext > 0.0f ? (c[k] - mn) / ext : 0.0f;
        float s = n * 2097152.0f;
        s = fminf(fmaxf(s, 0.0f), 2097151.0f);
        q[k] = (uint64_t)s;
    }
    keys[gid] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[gid] = gid;
}

// ------------------------------------------------------------------ LSD radix sort (8-bit digits)
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;                      // per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;

__global__ void __launch_bounds__(SORT_THREADS) k_sort_hist(const uint64_t *__restrict__ keys, uint32_t n, int shift, uint32_t nblocks, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_ITEMS; r++) {
        uint32_t i = base + r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    ghist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `count` uint32 in place, single workgroup of 1024 threads, four entries per thread and step
__global__ void __launch_bounds__(1024) k_scan_exclusive(uint32_t *__restrict__ data, uint32_t count, const uint32_t *__restrict__ items_dev /* count = ceil(*items_dev / 1024) */) {
    if (items_dev) count = (*items_dev + 1023u) >> 10;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t base = 0; base < count; base += 4096) {
        const uint32_t i = base + 4 * threadIdx.x;
        uint32_t v[4];
        for (int k = 0; k < 4; k++) v[k] = i + k < count ? data[i + k] : 0;
        const uint32_t s4 = v[0] + v[1] + v[2] + v[3];
        uint32_t x = s4;
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= (uint32_t)o) x += y; }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t k = 0; k < w; k++) woff += wsum[k];
        const uint32_t carry = carry_s;
        uint32_t e = carry + woff + x - s4;
        for (int k = 0; k < 4; k++) { if (i + k < count) data[i + k] = e; e += v[k]; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(SORT_THREADS) k_sort_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                               uint64_t *__restrict__ kout, uint32_t *__restrict__ vout,
                                                               const uint32_t *__restrict__ gofs, uint32_t n, int shift, uint32_t nblocks) {
    __shared__ uint32_t wcnt[4][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int k = 0; k < 4; k++) wcnt[k][tid] = 0;
    __syncthreads();
    const uint32_t wave_base = blockIdx.x * SORT_TILE + w * (SORT_ITEMS * 64);   // each wave owns a contiguous run: stable order = (wave, round, lane)
    uint64_t key[SORT_ITEMS];
    uint32_t off[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        uint32_t i = wave_base + r * 64 + lane;
        bool valid = i < n;
        key[r] = valid ? kin[i] : 0ull;
        uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
        uint32_t cnt = __popcll(peers);
        int leader = valid ? (__ffsll((long long)peers) - 1) : (int)lane;
        uint32_t base = 0;
        if (valid && (int)lane == leader) base = atomicAdd(&wcnt[w][d], cnt);
        base = __shfl(base, leader);
        off[r] = base + rank;
    }
    __syncthreads();
    {   // exclusive prefix over the 4 waves, per digit
        uint32_t c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid];
        wcnt[0][tid] = 0; wcnt[1][tid] = c0; wcnt[2][tid] = c0 + c1; wcnt[3][tid] = c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        uint32_t i = wave_base + r * 64 + lane;
        if (i < n) {
            uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            uint32_t pos = gofs[d * nblocks + blockIdx.x] + wcnt[w][d] + off[r];
            kout[pos] = key[r];
            vout[pos] = vin[i];
        }
    }
}

// ------------------------------------------------------------------ Karras 2012 radix tree
__device__ __forceinline__ int delta_kr(const uint64_t *__restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}
// node ids: internal i -> i (0..n-2), leaf j -> (n-1)+j
__global__ void k_karras(const uint64_t *__restrict__ keys, int n, uint32_t *__restrict__ left, uint32_t *__restrict__ right, uint32_t *__restrict__ parent) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta_kr(keys, n, i, i + 1) - delta_kr(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta_kr(keys, n, i, i - d);
    int lmax = 2;
    while (delta_kr(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta_kr(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta_kr(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta_kr(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    uint32_t lc = (min(i, j) == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    uint32_t rc = (max(i, j) == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    left[i] = lc; right[i] = rc;
    parent[lc] = (uint32_t)i; parent[rc] = (uint32_t)i;
    if (i == 0) parent[0] = NONE;
}

// ------------------------------------------------------------------ refit + SAH collapse
struct TreeArrays {
    float4 *lo, *hi;          // per node (2n-1)
    uint32_t *parent;         // per node
    uint32_t *left, *right;   // per internal node id (size n-1, indexed by node id < n-1 for Karras; PLOC uses ids too)
    uint32_t *flags;          // per node, zeroed
    float *cost;              // per node
    uint32_t *ntri, *size;    // per node: triangles below, surviving nodes below (incl. self)
    uint8_t *collapsed;       // per node: 1 = becomes a leaf of the emitted tree
    uint8_t *mask;            // per node: near-child mask
};

struct WideDP { float *C; uint8_t *dec; };      // optimal 8-wide collapse (k_wide_dp below): 8 entries per node; C == nullptr: not wanted

__device__ __forceinline__ float box_area(float4 lo, float4 hi) {
    float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// one node of the optimal 8-wide collapse (see "optimal 8-wide collapse" below): C(p, .) from the two children's entries cl[1..7], cr[1..7]
__device__ __forceinline__ void wide_dp_node(const float *cl, const float *cr, float *C, uint8_t *D, float area, uint32_t nt, int max_leaf, float c_node, float c_tri) {
    float best = 3.0e38f; int bk = 1;
    for (int k = 1; k <= WIDE_N - 1; k++) { const float c = cl[k] + cr[WIDE_N - k]; if (c < best) { best = c; bk = k; } }
    const float c_int = area * c_node + best;
    const float c_leaf = nt <= (uint32_t)max_leaf ? area * (float)nt * c_tri : 3.0e38f;
    D[0] = (uint8_t)bk; D[1] = c_leaf <= c_int ? 1 : 0; C[1] = fminf(c_leaf, c_int); C[0] = c_int;
    for (int i = 2; i <= WIDE_N - 1; i++) {
        float b = 3.0e38f; int k_ = 1;
        for (int k = 1; k < i; k++) { const float c = cl[k] + cr[i - k]; if (c < b) { b = c; k_ = k; } }
        if (b < C[i - 1]) { C[i] = b; D[i] = (uint8_t)k_; } else { C[i] = C[i - 1]; D[i] = 0; }
    }
}

// Bottom-up pass: one thread per leaf climbs; at every node the thread that arrives second computes it from the two children.  What one thread hands to
// another (possibly on another XCD, whose L2 is not coherent with this one's) are the child's box, {cost, triangles, size} and its C(., 1..7): these are
// written with 16-byte write-through stores (sc1), drained (s_waitcnt vmcnt(0)) before the agent-scope atomic on the parent's arrival counter, and read by
// the last arriver with sc1 loads issued after its atomic has returned (MI355X_MICROARCH.md "Valid forms": every handed-off byte stored sc1 and drained
// before the counter, every load of them an sc1 load to registers).  The first version used two __threadfence() per level instead — an L2 write-back and
// an L1 invalidate, ~3.5 us each, from every climbing thread: 8.8 of the build's 21.8 ms for 885 K triangles; this one takes 0.9 ms.
typedef unsigned int refit_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t refit_rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)0xFFFFFFF0u, 0x00020000); }
__device__ __forceinline__ float4 refit_ld_wt(__amdgpu_buffer_rsrc_t r, uint32_t index) {
    const refit_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, index * 16u, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void refit_st_wt(__amdgpu_buffer_rsrc_t r, uint32_t index, float4 a) {
    refit_u32x4 v; v.x = __float_as_uint(a.x); v.y = __float_as_uint(a.y); v.z = __float_as_uint(a.z); v.w = __float_as_uint(a.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, index * 16u, 0, 16);
}

// FENCED (scene option refit_fenced = 1): the same pass with the textbook hand-off — plain stores, __threadfence() before the arrival is counted and after it — instead of write-through
// stores, s_waitcnt and sc1 loads.  8.8 ms instead of 0.3 for 885 K triangles; kept as the reference the fast form is compared with bit for bit (tests/test_build_sizes.py).
template <bool FENCED>
__global__ void k_refit(TreeArrays t, const uint32_t *__restrict__ vals, const float4 *__restrict__ leaf_lo, const float4 *__restrict__ leaf_hi,
                        uint32_t n, uint32_t leaf_base, int max_leaf, float ct, float ci, float4 *__restrict__ aux /* per node {cost, triangles, size, -} */,
                        WideDP dp, int dp_max_leaf, float dp_c_node, float dp_c_tri) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const __amdgpu_buffer_rsrc_t r_lo = refit_rsrc(t.lo), r_hi = refit_rsrc(t.hi), r_aux = refit_rsrc(aux), r_C = refit_rsrc(dp.C);
    float4 *const p_lo = t.lo, *const p_hi = t.hi, *const p_aux = aux, *const p_C = reinterpret_cast<float4 *>(dp.C);
#define RST(name, idx, val) do { if (FENCED) p_##name[idx] = (val); else refit_st_wt(r_##name, idx, val); } while (0)
#define RLD(name, idx) (FENCED ? p_##name[idx] : refit_ld_wt(r_##name, idx))
    uint32_t node = leaf_base + j;
    uint32_t gid = vals[j];
    float4 lo = leaf_lo[gid], hi = leaf_hi[gid];
    RST(lo, node, lo); RST(hi, node, hi);
    const float leaf_cost = ci * box_area(lo, hi);
    RST(aux, node, make_float4(leaf_cost, __uint_as_float(1u), __uint_as_float(1u), 0.0f));
    t.cost[node] = leaf_cost;
    t.ntri[node] = 1; t.size[node] = 1; t.collapsed[node] = 1; t.mask[node] = 0;
    if (dp.C) {
        const float c = dp_c_tri * box_area(lo, hi);
        RST(C, 2 * node, make_float4(c, c, c, c)); RST(C, 2 * node + 1, make_float4(c, c, c, c));
        for (int i = 0; i < 8; i++) dp.dec[8 * (size_t)node + i] = i == 1 ? 1 : 0;
    }
    uint32_t p = t.parent[node];
    while (p != NONE) {
        if (FENCED) __threadfence(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this node's stores have left the CU ...
        const uint32_t old = __hip_atomic_fetch_add(&t.flags[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ... before the arrival is counted
        if (old == 0) return;                       // the sibling subtree finishes this node
        if (FENCED) __threadfence(); else asm volatile("" ::: "memory");              // the children are read after the arrival has been counted, not before
        uint32_t l = t.left[p], r = t.right[p];
        const float4 llo = RLD(lo, l), lhi = RLD(hi, l), rlo = RLD(lo, r), rhi = RLD(hi, r);
        const float4 la = RLD(aux, l), ra = RLD(aux, r);
        float4 blo = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), 0.0f);
        float4 bhi = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.0f);
        float area = box_area(blo, bhi);
        uint32_t nt = __float_as_uint(la.y) + __float_as_uint(ra.y);
        float c_inner = ct * area + la.x + ra.x;
        float c_leaf = ci * area * (float)nt;
        bool col = (nt <= (uint32_t)max_leaf) && (c_leaf <= c_inner);
        // near-child mask: along the axis where the child centres are furthest apart, the child with
        // the smaller centre is entered first by rays travelling in +axis.
        float cl[3] = {llo.x + lhi.x, llo.y + lhi.y, llo.z + lhi.z}, cr[3] = {rlo.x + rhi.x, rlo.y + rhi.y, rlo.z + rhi.z};
        int ax = 0; float best = fabsf(cl[0] - cr[0]);
        if (fabsf(cl[1] - cr[1]) > best) { best = fabsf(cl[1] - cr[1]); ax = 1; }
        if (fabsf(cl[2] - cr[2]) > best) { ax = 2; }
        bool left_greater = cl[ax] > cr[ax];
        uint32_t m = 0;
        for (int o = 0; o < 8; o++) { bool negdir = (o >> ax) & 1; if (negdir != left_greater) m |= 1u << o; }
        const float cost = col ? c_leaf : c_inner;
        const uint32_t size = col ? 1u : 1u + __float_as_uint(la.z) + __float_as_uint(ra.z);
        RST(lo, p, blo); RST(hi, p, bhi);
        RST(aux, p, make_float4(cost, __uint_as_float(nt), __uint_as_float(size), 0.0f));
        t.cost[p] = cost;
        t.ntri[p] = nt;
        t.size[p] = size;
        t.collapsed[p] = col ? 1 : 0;
        t.mask[p] = (uint8_t)m;
        if (dp.C) {                                 // same bottom-up pass: the children's entries are complete
            const float4 l0 = RLD(C, 2 * l), l1 = RLD(C, 2 * l + 1), r0 = RLD(C, 2 * r), r1 = RLD(C, 2 * r + 1);
            const float dl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w}, dr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            float C[8]; uint8_t D[8];
            wide_dp_node(dl, dr, C, D, area, nt, dp_max_leaf, dp_c_node, dp_c_tri);
            RST(C, 2 * p, make_float4(C[0], C[1], C[2], C[3])); RST(C, 2 * p + 1, make_float4(C[4], C[5], C[6], C[7]));
            for (int i = 0; i < 8; i++) dp.dec[8 * (size_t)p + i] = D[i];
        }
        node = p;
        p = t.parent[p];
    }
}
#undef RST
#undef RLD

// ------------------------------------------------------------------ numbering + emit
__global__ void k_assign(TreeArrays t, uint32_t nnodes, uint32_t *__restrict__ new_index, uint32_t *__restrict__ leaf_offset, uint32_t *__restrict__ stat /*[0]=max depth,[1]=leaves*/) {
    uint32_t nd = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t my_depth = 0, my_leaf = 0;
    if (nd < nnodes) {
        bool dropped = false;
        uint32_t pre = 0, lrank = 0, depth = 0, c = nd, p;
        while ((p = t.parent[c]) != NONE) {
            if (t.collapsed[p]) dropped = true;
            uint32_t l = t.left[p];
            if (l != c) { pre += t.size[l]; lrank += t.ntri[l]; }
            pre += 1; depth++; c = p;
        }
        new_index[nd] = dropped ? NONE : pre;
        leaf_offset[nd] = lrank;
        if (depth == 0) stat[2] = nd;               // the root (the one node without a parent)
        if (!dropped) { my_depth = depth; my_leaf = t.collapsed[nd] ? 1u : 0u; }
    }
    // depth and leaf count: one pair of atomics per workgroup (one per wave was this kernel's whole duration)
    __shared__ uint32_t sd[16], sl[16];
    for (int o = 32; o > 0; o >>= 1) { my_depth = max(my_depth, (uint32_t)__shfl_xor((int)my_depth, o)); my_leaf += (uint32_t)__shfl_xor((int)my_leaf, o); }
    if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = my_depth; sl[threadIdx.x >> 6] = my_leaf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t d = 0, l = 0;
        for (uint32_t i = 0; i < ((blockDim.x + 63) >> 6); i++) { d = max(d, sd[i]); l += sl[i]; }
        atomicMax(&stat[0], d);
        if (l) atomicAdd(&stat[1], l);
    }
}

// what the host wants to know about the finished tree, in one 48-byte read-back: {root, size, cost, depth, leaves, -, lo.xyz, hi.xyz}
__global__ void k_build_summary(TreeArrays t, const uint32_t *__restrict__ stat, uint32_t *__restrict__ out) {
    const uint32_t root = stat[2];
    if (root == NONE) { out[0] = NONE; return; }
    out[0] = root; out[1] = t.size[root]; out[2] = __float_as_uint(t.cost[root]); out[3] = stat[0]; out[4] = stat[1]; out[5] = 0;
    const float4 lo = t.lo[root], hi = t.hi[root];
    out[6] = __float_as_uint(lo.x); out[7] = __float_as_uint(lo.y); out[8] = __float_as_uint(lo.z);
    out[9] = __float_as_uint(hi.x); out[10] = __float_as_uint(hi.y); out[11] = __float_as_uint(hi.z);
}

__global__ void k_emit_nodes(TreeArrays t, uint32_t nnodes, const uint32_t *__restrict__ new_index, const uint32_t *__restrict__ leaf_offset, float4 *__restrict__ out) {
    uint32_t nd = blockIdx.x * blockDim.x + threadIdx.x;
    if (nd >= nnodes) return;
    uint32_t me = new_index[nd];
    if (me == NONE) return;
    uint32_t esc[8];
#pragma unroll
    for (int o = 0; o < 8; o++) {
        uint32_t c = nd, e = NODE_TERM;
        for (;;) {
            uint32_t p = t.parent[c];
            if (p == NONE) break;
            uint32_t l = t.left[p], r = t.right[p];
            uint32_t near = ((t.mask[p] >> o) & 1u) ? r : l;
            if (c == near) { e = new_index[near == l ? r : l]; break; }
            c = p;
        }
        esc[o] = e;
    }
    float4 lo = t.lo[nd], hi = t.hi[nd];
    uint32_t a, b;
    if (t.collapsed[nd]) { a = NODE_LEAF | leaf_offset[nd]; b = t.ntri[nd]; }
    else { a = new_index[t.left[nd]]; b = ((uint32_t)t.mask[nd] << 24) | new_index[t.right[nd]]; }
    out[4 * (size_t)me + 0] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(a));
    out[4 * (size_t)me + 1] = make_float4(hi.x, hi.y, hi.z, __uint_as_float(b));
    out[4 * (size_t)me + 2] = make_float4(__uint_as_float(esc[0]), __uint_as_float(esc[1]), __uint_as_float(esc[2]), __uint_as_float(esc[3]));
    out[4 * (size_t)me + 3] = make_float4(__uint_as_float(esc[4]), __uint_as_float(esc[5]), __uint_as_float(esc[6]), __uint_as_float(esc[7]));
}

__global__ void k_emit_packets(const uint32_t *__restrict__ vals, const uint32_t *__restrict__ leaf_offset, uint32_t leaf_base, uint32_t n,
                               const float4 *__restrict__ tri_world, const uint32_t *__restrict__ ref_tri /* nullptr: reference r is triangle r */, float4 *__restrict__ packets) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t rank = leaf_offset[leaf_base + j];
    uint32_t gid = ref_tri ? ref_tri[vals[j]] : vals[j];
    packets[3 * (size_t)rank + 0] = tri_world[3 * (size_t)gid + 0];
    packets[3 * (size_t)rank + 1] = tri_world[3 * (size_t)gid + 1];
    packets[3 * (size_t)rank + 2] = tri_world[3 * (size_t)gid + 2];
}

// ------------------------------------------------------------------ PLOC (builder 1)
// Clusters live in Morton order.  Each round: every cluster finds, within +-radius positions, the
// neighbour minimising the surface area of the merged box; mutual pairs merge (the lower position
// creates the node); survivors are compacted, order preserved.  Node ids: leaves (n-1)+j as for
// Karras, internal nodes allocated from an atomic counter 0..n-2; the last one created is the root.
__global__ void k_ploc_init(uint32_t n, uint32_t leaf_base, const uint32_t *__restrict__ vals, const float4 *__restrict__ leaf_lo, const float4 *__restrict__ leaf_hi,
                            uint32_t *__restrict__ cid, float4 *__restrict__ clo, float4 *__restrict__ chi) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t gid = vals[j];
    cid[j] = leaf_base + j; clo[j] = leaf_lo[gid]; chi[j] = leaf_hi[gid];
}

// (m_dev, here and below: the cluster count of this round as the previous round left it on the device — the host launches a few rounds on its last known count,
// an upper bound, and reads the count back once per batch instead of once per round)
__global__ void k_ploc_nn(uint32_t m, int radius, const float4 *__restrict__ clo, const float4 *__restrict__ chi, uint32_t *__restrict__ nn, const uint32_t *__restrict__ m_dev) {
    if (m_dev) m = *m_dev;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    float4 lo = clo[i], hi = chi[i];
    float best = 3.0e38f; uint32_t bj = NONE;
    int j0 = (int)i - radius; if (j0 < 0) j0 = 0;
    int j1 = (int)i + radius; if (j1 > (int)m - 1) j1 = (int)m - 1;
    for (int j = j0; j <= j1; j++) {
        if (j == (int)i) continue;
        float4 l2 = clo[j], h2 = chi[j];
        float dx = fmaxf(hi.x, h2.x) - fminf(lo.x, l2.x), dy = fmaxf(hi.y, h2.y) - fminf(lo.y, l2.y), dz = fmaxf(hi.z, h2.z) - fminf(lo.z, l2.z);
        float a = dx * dy + dy * dz + dz * dx;
        if (a < best) { best = a; bj = (uint32_t)j; }      // ties: lowest position wins (deterministic)
    }
    nn[i] = bj;
}

__global__ void k_ploc_merge(uint32_t m, const uint32_t *__restrict__ nn, const uint32_t *__restrict__ cid, const float4 *__restrict__ clo, const float4 *__restrict__ chi,
                             uint32_t *__restrict__ keep /* 1 = survives (possibly as merged) */, uint32_t *__restrict__ new_cid, float4 *__restrict__ nlo, float4 *__restrict__ nhi,
                             uint32_t *__restrict__ node_counter, uint32_t *__restrict__ left, uint32_t *__restrict__ right, uint32_t *__restrict__ parent,
                             float4 *__restrict__ node_lo, float4 *__restrict__ node_hi, const uint32_t *__restrict__ m_dev) {
    if (m_dev) m = *m_dev;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = i < m ? nn[i] : NONE;
    const bool mutual = (j != NONE) && (nn[j] == i);
    // node ids: one atomic per workgroup (one per wave — 14 K in the first round — was this kernel's duration), then rank within the workgroup
    __shared__ uint32_t wcnt[16], wbase[16];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long creators = __ballot(mutual && i < j);
    if (lane == 0) wcnt[wv] = (uint32_t)__popcll(creators);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t k = 0; k < ((blockDim.x + 63) >> 6); k++) { wbase[k] = tot; tot += wcnt[k]; }
        const uint32_t base = tot ? atomicAdd(node_counter, tot) : 0u;
        for (uint32_t k = 0; k < ((blockDim.x + 63) >> 6); k++) wbase[k] += base;
    }
    __syncthreads();
    if (i >= m) return;
    if (mutual && i < j) {
        uint32_t id = wbase[wv] + (uint32_t)__popcll(creators & ((1ull << lane) - 1ull));
        uint32_t a = cid[i], b = cid[j];
        left[id] = a; right[id] = b; parent[a] = id; parent[b] = id;
        float4 lo = make_float4(fminf(clo[i].x, clo[j].x), fminf(clo[i].y, clo[j].y), fminf(clo[i].z, clo[j].z), 0.0f);
        float4 hi = make_float4(fmaxf(chi[i].x, chi[j].x), fmaxf(chi[i].y, chi[j].y), fmaxf(chi[i].z, chi[j].z), 0.0f);
        node_lo[id] = lo; node_hi[id] = hi;
        keep[i] = 1; new_cid[i] = id; nlo[i] = lo; nhi[i] = hi;
    } else if (mutual) {
        keep[i] = 0;
    } else {
        keep[i] = 1; new_cid[i] = cid[i]; nlo[i] = clo[i]; nhi[i] = chi[i];
    }
}

__global__ void k_ploc_compact(uint32_t m, const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos /* exclusive scan of keep */,
                               const uint32_t *__restrict__ new_cid, const float4 *__restrict__ nlo, const float4 *__restrict__ nhi,
                               uint32_t *__restrict__ cid, float4 *__restrict__ clo, float4 *__restrict__ chi, const uint32_t *__restrict__ m_dev) {
    if (m_dev) m = *m_dev;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (keep[i]) { uint32_t p = pos[i]; cid[p] = new_cid[i]; clo[p] = nlo[i]; chi[p] = nhi[i]; }
}

// multi-block exclusive scan: per-block sums -> scan -> add
__global__ void __launch_bounds__(1024) k_scan_block(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *__restrict__ bsum, uint32_t count, const uint32_t *__restrict__ count_dev) {
    if (count_dev) count = *count_dev;
    __shared__ uint32_t wsum[16];
    uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t v = i < count ? in[i] : 0, x = v;
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= (uint32_t)o) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (uint32_t k = 0; k < 16; k++) { if (k < w) woff += wsum[k]; tot += wsum[k]; }
    if (i < count) out[i] = woff + x - v;
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ void k_scan_add(uint32_t *__restrict__ out, const uint32_t *__restrict__ bsum_scanned, uint32_t count, uint32_t *__restrict__ total, const uint32_t *__restrict__ last_in, const uint32_t *__restrict__ count_dev) {
    if (count_dev) count = *count_dev;
    uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i < count) {
        uint32_t v = out[i] + bsum_scanned[blockIdx.x];
        out[i] = v;
        if (i == count - 1) *total = v + last_in[i];
    }
}
__global__ void k_set_root_parent(const uint32_t *__restrict__ cid, uint32_t *__restrict__ parent) { parent[cid[0]] = NONE; }
// The last rounds of PLOC in ONE workgroup: once at most PLOC_TAIL clusters are left (about 90 of the 106 rounds of an 885 K-triangle build, 6 launches and one
// host read-back each) the clusters fit in LDS and the rounds need a barrier, not a launch.  Same rounds, same pairs, same order as the kernels above (node ids are
// handed out in a different order; the numbering of the emitted tree does not depend on them).
constexpr uint32_t PLOC_TAIL = 1024;
constexpr int PLOC_ROUNDS_PER_READBACK = 4;     // a round removes at most half the clusters: more than 1024 / 16 are left when a batch ends
__global__ void __launch_bounds__(1024) k_ploc_tail(uint32_t m, int radius, const uint32_t *__restrict__ cid, const float4 *__restrict__ clo, const float4 *__restrict__ chi,
                                                    uint32_t *__restrict__ node_counter, uint32_t *__restrict__ left, uint32_t *__restrict__ right, uint32_t *__restrict__ parent,
                                                    float4 *__restrict__ node_lo, float4 *__restrict__ node_hi) {
    __shared__ float4 slo[PLOC_TAIL], shi[PLOC_TAIL];
    __shared__ uint32_t scid[PLOC_TAIL], snn[PLOC_TAIL], wsum[16], s_next;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < m) { slo[tid] = clo[tid]; shi[tid] = chi[tid]; scid[tid] = cid[tid]; }
    if (tid == 0) s_next = *node_counter;
    __syncthreads();
    while (m > 1) {
        if (tid < m) {
            const float4 lo = slo[tid], hi = shi[tid];
            float best = 3.0e38f; uint32_t bj = NONE;
            int j0 = (int)tid - radius; if (j0 < 0) j0 = 0;
            int j1 = (int)tid + radius; if (j1 > (int)m - 1) j1 = (int)m - 1;
            for (int j = j0; j <= j1; j++) {
                if (j == (int)tid) continue;
                const float4 l2 = slo[j], h2 = shi[j];
                float dx = fmaxf(hi.x, h2.x) - fminf(lo.x, l2.x), dy = fmaxf(hi.y, h2.y) - fminf(lo.y, l2.y), dz = fmaxf(hi.z, h2.z) - fminf(lo.z, l2.z);
                float a = dx * dy + dy * dz + dz * dx;
                if (a < best) { best = a; bj = (uint32_t)j; }
            }
            snn[tid] = bj;
        }
        __syncthreads();
        uint32_t keep = 0, ncid = 0; float4 nlo = make_float4(0, 0, 0, 0), nhi = nlo;
        if (tid < m) {
            const uint32_t j = snn[tid];
            const bool mutual = (j != NONE) && (snn[j] == tid);
            if (mutual && tid < j) {
                const uint32_t id = atomicAdd(&s_next, 1u);
                const uint32_t a = scid[tid], b = scid[j];
                left[id] = a; right[id] = b; parent[a] = id; parent[b] = id;
                nlo = make_float4(fminf(slo[tid].x, slo[j].x), fminf(slo[tid].y, slo[j].y), fminf(slo[tid].z, slo[j].z), 0.0f);
                nhi = make_float4(fmaxf(shi[tid].x, shi[j].x), fmaxf(shi[tid].y, shi[j].y), fmaxf(shi[tid].z, shi[j].z), 0.0f);
                node_lo[id] = nlo; node_hi[id] = nhi;
                keep = 1; ncid = id;
            } else if (!mutual) {
                keep = 1; ncid = scid[tid]; nlo = slo[tid]; nhi = shi[tid];
            }
        }
        uint32_t x = keep;
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= (uint32_t)o) x += y; }
        if (lane == 63) wsum[w] = x;
        __syncthreads();                            // every read of this round's clusters is done
        uint32_t woff = 0, tot = 0;
        for (uint32_t k = 0; k < 16; k++) { if (k < w) woff += wsum[k]; tot += wsum[k]; }
        if (tot >= m) { if (tid == 0) *node_counter = NONE; return; }      // no pair merged (boxes that compare false with everything): the host reports it
        if (keep) { const uint32_t pos = woff + x - 1u; scid[pos] = ncid; slo[pos] = nlo; shi[pos] = nhi; }
        m = tot;
        __syncthreads();
    }
    if (tid == 0) { parent[scid[0]] = NONE; *node_counter = s_next; }
}

// ------------------------------------------------------------------ optimal 8-wide collapse (wide_collapse = 1)
// After Ylitie, Karras, Laine 2017 ("Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", §3.1): for every node n of the
// binary tree, C(n, i) = the least SAH cost of representing n's subtree as a forest of at most i wide-BVH roots, i = 1..7:
//   C(n, 1)  = min(C_leaf(n), C_node(n)),   C_leaf = area * triangles * c_tri  (subtrees of <= max_leaf triangles),
//                                           C_node = area * c_node + min_k C(left, k) + C(right, 8 - k)
//   C(n, i)  = min(C(n, i - 1), min_k C(left, k) + C(right, i - k))
// computed bottom-up inside k_refit (wide_dp_node: one pass over the tree for boxes, SAH leaf collapse and this table), with the arg-min of every entry kept: dec[0] = k of C_node, dec[1] = 1 when the
// leaf is cheaper, dec[i] = k of the split or 0 for "no better than i - 1 roots".  k_wide_level<true> then unfolds the decisions of a
// wide node's root into its (at most eight) children.  The greedy collapse it replaces opens the largest child first and fills 6.0 of 8
// slots on DragonScene; this one trades nodes against triangle tests with the constants the traversal kernel was measured at.
// ------------------------------------------------------------------ 8-wide collapse + quantisation
// One thread per wide node of the current level.  Greedy collapse (largest surface area first) of the
// refitted binary tree; children that are binary inner nodes form the next level (BFS numbering, so a
// node's children are contiguous and the top of the tree sits at the lowest indices).
// Levels are launched in batches without a host round trip in between (the 13 levels of DragonScene used to cost 13 read-backs, 0.9 of the build's 3.9 ms): level L takes its
// node count from lv[L] — lv[0] = 1, level L - 1 has counted its internal children into lv[L] — and its first node index from the counts before it; the grid is sized for
// the most a level can hold (min(8^L, max_w)), surplus threads leave.  lv[WIDE_LV_PACKETS] counts packets, lv[WIDE_LV_ERROR] is set when the node array would overflow.
constexpr uint32_t WIDE_LV_MAX = 4096, WIDE_LV_PACKETS = WIDE_LV_MAX + 1, WIDE_LV_ERROR = WIDE_LV_MAX + 2, WIDE_LV_WORDS = WIDE_LV_MAX + 3;
template <bool DP>
__global__ void k_wide_level(TreeArrays t, WideDP dp, const uint32_t *__restrict__ leaf_offset, const uint32_t *__restrict__ fin, uint32_t level, uint32_t max_w,
                             uint32_t *__restrict__ fout, uint32_t *__restrict__ lv,
                             float4 *__restrict__ wnodes, const float4 *__restrict__ packets, float4 *__restrict__ wpackets) {
    const uint32_t n_in = lv[level];
    if (blockIdx.x * blockDim.x >= n_in) return;
    uint32_t base_in = 0;
    for (uint32_t l = 0; l < level; l++) base_in += lv[l];
    const uint32_t next_base = base_in + n_in;
    uint32_t *const counters = lv + level + 1;            // [0]: the next level's nodes
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n_in;
    uint32_t ch[8]; bool isleaf[8]; int nch = 0;
    uint32_t ninner = 0, ntris = 0;
    float4 nlo = make_float4(0, 0, 0, 0), nhi = nlo;
    if (active && DP) {
        const uint32_t f = fin[i];
        nlo = t.lo[f]; nhi = t.hi[f];
        if (base_in == 0 && dp.dec[8 * (size_t)f + 1]) { ch[0] = f; isleaf[0] = true; nch = 1; }     // the whole scene is one leaf
        else {
            // unfold the decisions: (node, roots allowed) pairs, depth first; at most eight children come out
            uint32_t st_n[8]; uint8_t st_i[8]; int sp = 0;
            { const uint32_t k = dp.dec[8 * (size_t)f + 0]; st_n[sp] = t.right[f]; st_i[sp++] = (uint8_t)((uint32_t)WIDE_N - k); st_n[sp] = t.left[f]; st_i[sp++] = (uint8_t)k; }
            while (sp > 0) {
                const uint32_t c = st_n[--sp]; uint32_t b = st_i[sp];
                while (b > 1u && dp.dec[8 * (size_t)c + b] == 0) b--;
                if (b == 1u) { ch[nch] = c; isleaf[nch] = dp.dec[8 * (size_t)c + 1] != 0; nch++; }
                else { const uint32_t k = dp.dec[8 * (size_t)c + b]; st_n[sp] = t.right[c]; st_i[sp++] = (uint8_t)(b - k); st_n[sp] = t.left[c]; st_i[sp++] = (uint8_t)k; }
            }
        }
        for (int k = 0; k < nch; k++) { if (isleaf[k]) ntris += t.ntri[ch[k]]; else ninner++; }
    }
    if (active && !DP) {
        const uint32_t f = fin[i];
        nlo = t.lo[f]; nhi = t.hi[f];
        if (t.collapsed[f]) { ch[0] = f; isleaf[0] = true; nch = 1; }
        else {
            ch[0] = t.left[f]; ch[1] = t.right[f]; nch = 2;
            isleaf[0] = t.collapsed[ch[0]]; isleaf[1] = t.collapsed[ch[1]];
            // phase 1: open inner children, largest surface area first
            while (nch < WIDE_N) {
                int best = -1; float ba = -1.0f;
                for (int k = 0; k < nch; k++)
                    if (!isleaf[k]) { float a = box_area(t.lo[ch[k]], t.hi[ch[k]]); if (a > ba) { ba = a; best = k; } }
                if (best < 0) break;
                uint32_t c = ch[best];
                ch[best] = t.left[c]; isleaf[best] = t.collapsed[ch[best]];
                ch[nch] = t.right[c]; isleaf[nch] = t.collapsed[ch[nch]]; nch++;
            }
        }
        // phase 2: slots left over are free box tests — split multi-triangle leaves back along the binary tree, so
        // that fewer triangles (one sequential round trip each) are tested behind every box that is hit
        while (nch < WIDE_N) {
            int best = -1; float ba = -1.0f;
            for (int k = 0; k < nch; k++)
                if (isleaf[k] && t.ntri[ch[k]] > 1) { float a = box_area(t.lo[ch[k]], t.hi[ch[k]]); if (a > ba) { ba = a; best = k; } }
            if (best < 0) break;
            uint32_t c = ch[best];
            ch[best] = t.left[c]; ch[nch] = t.right[c]; isleaf[nch] = true; nch++;
        }
        for (int k = 0; k < nch; k++) { if (isleaf[k]) ntris += t.ntri[ch[k]]; else ninner++; }
    }
    // wave-aggregated reservation of next-level node slots and packet slots
    const uint32_t lane = threadIdx.x & 63;
    uint32_t xi = ninner, xt = ntris;
    for (int o = 1; o < 64; o <<= 1) { uint32_t a = __shfl_up(xi, o), b = __shfl_up(xt, o); if (lane >= (uint32_t)o) { xi += a; xt += b; } }
    uint32_t tot_i = __shfl(xi, 63), tot_t = __shfl(xt, 63), base_i = 0, base_t = 0;
    if (lane == 63) { if (tot_i) base_i = atomicAdd(&counters[0], tot_i); if (tot_t) base_t = atomicAdd(&lv[WIDE_LV_PACKETS], tot_t); }
    base_i = __shfl(base_i, 63); base_t = __shfl(base_t, 63);
    if (!active) return;
    if ((uint64_t)next_base + base_i + tot_i > (uint64_t)max_w) { lv[WIDE_LV_ERROR] = 1u; return; }      // (wave-uniform) the node array would overflow: the host reports it
    const uint32_t my_i = base_i + xi - ninner, my_t = base_t + xt - ntris;
    // slot assignment: preferred slot = side of the node centre per axis; greedy nearest free slot (Hamming)
    int slot_of[8]; bool used[8] = {false, false, false, false, false, false, false, false};
    const float cx = nlo.x + nhi.x, cy = nlo.y + nhi.y, cz = nlo.z + nhi.z;      // 2 x centre
    for (int k = 0; k < nch; k++) {
        uint32_t c = ch[k];
        float4 lo = t.lo[c], hi = t.hi[c];
        int pref = ((lo.x + hi.x) > cx ? 1 : 0) | ((lo.y + hi.y) > cy ? 2 : 0) | ((lo.z + hi.z) > cz ? 4 : 0);
        int bs = -1, bd = 99;
        for (int sl = 0; sl < 8; sl++) if (!used[sl]) { int dd = __popc((unsigned)(sl ^ pref)); if (dd < bd) { bd = dd; bs = sl; } }
        used[bs] = true; slot_of[k] = bs;
    }
    int child_in_slot[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    for (int k = 0; k < nch; k++) child_in_slot[slot_of[k]] = k;
    // quantisation grid: p = lo, step 2^e >= extent / 255 per axis
    float ext[3] = {nhi.x - nlo.x, nhi.y - nlo.y, nhi.z - nlo.z};
    uint32_t eb[3]; float inv_step[3], step[3];
    for (int a = 0; a < 3; a++) {
        float sdiv = ext[a] / 255.0f;
        uint32_t bits = __float_as_uint(sdiv);
        uint32_t e = (bits >> 23) + ((bits & 0x7FFFFFu) ? 1u : 0u);
        if (e < 1u) e = 1u; if (e > 254u) e = 254u;
        eb[a] = e;
        step[a] = __uint_as_float(e << 23);
        inv_step[a] = __uint_as_float((254u - e) << 23);        // 2^-(e-127)
    }
    const float pl[3] = {nlo.x, nlo.y, nlo.z};
    uint32_t q[6][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};     // qlo x,y,z then qhi x,y,z; 2 dwords (8 bytes) each
    uint32_t meta[2] = {0, 0}, imask = 0, rank_i = 0, off_t = 0;
    for (int sl = 0; sl < 8; sl++) {
        int k = child_in_slot[sl];
        uint32_t ql[3] = {255, 255, 255}, qh[3] = {0, 0, 0};
        if (k >= 0) {
            uint32_t c = ch[k];
            float4 lo = t.lo[c], hi = t.hi[c];
            const float cl[3] = {lo.x, lo.y, lo.z}, chh[3] = {hi.x, hi.y, hi.z};
            for (int a = 0; a < 3; a++) {
                float fl = floorf((cl[a] - pl[a]) * inv_step[a]);
                float fh = ceilf((chh[a] - pl[a]) * inv_step[a]);
                fl = fminf(fmaxf(fl, 0.0f), 255.0f); fh = fminf(fmaxf(fh, 0.0f), 255.0f);
                // make sure the decoded planes still enclose the child after rounding of (c - p)
                if (pl[a] + fl * step[a] > cl[a] && fl > 0.0f) fl -= 1.0f;
                if (pl[a] + fh * step[a] < chh[a] && fh < 255.0f) fh += 1.0f;
                ql[a] = (uint32_t)fl; qh[a] = (uint32_t)fh;
            }
            if (isleaf[k]) {
                uint32_t cnt = t.ntri[c];
                uint32_t m = (cnt << 5) | off_t;
                meta[sl >> 2] |= m << (8 * (sl & 3));
                uint32_t src = leaf_offset[c];
                for (uint32_t r = 0; r < cnt; r++)
                    for (int j = 0; j < 3; j++) wpackets[WPK * (size_t)(my_t + off_t + r) + j] = packets[3 * (size_t)(src + r) + j];
                off_t += cnt;
            } else {
                imask |= 1u << sl;
                fout[my_i + rank_i] = c;
                rank_i++;
            }
        }
        for (int a = 0; a < 3; a++) { q[a][sl >> 2] |= ql[a] << (8 * (sl & 3)); q[3 + a][sl >> 2] |= qh[a] << (8 * (sl & 3)); }
    }
    const size_t w = WNODE_STRIDE * (size_t)(base_in + i);
    // exponents are stored unbiased (int8, e - 127): the traversal scales 1/direction with v_ldexp_f32
    wnodes[w + 0] = make_float4(nlo.x, nlo.y, nlo.z, __uint_as_float(((eb[0] - 127u) & 0xFFu) | (((eb[1] - 127u) & 0xFFu) << 8) | (((eb[2] - 127u) & 0xFFu) << 16) | (imask << 24)));
    wnodes[w + 1] = make_float4(__uint_as_float(next_base + my_i), __uint_as_float(my_t), __uint_as_float(meta[0]), __uint_as_float(meta[1]));
    wnodes[w + 2] = make_float4(__uint_as_float(q[0][0]), __uint_as_float(q[0][1]), __uint_as_float(q[1][0]), __uint_as_float(q[1][1]));
    wnodes[w + 3] = make_float4(__uint_as_float(q[2][0]), __uint_as_float(q[2][1]), __uint_as_float(q[3][0]), __uint_as_float(q[3][1]));
    wnodes[w + 4] = make_float4(__uint_as_float(q[4][0]), __uint_as_float(q[4][1]), __uint_as_float(q[5][0]), __uint_as_float(q[5][1]));
}

// ------------------------------------------------------------------ refit of the 8-wide layout (deformed geometry, same topology: mrt_scene_update_mesh + commit)
// The reference rebuilds nothing per frame (Renderer.swift:184-214 runs once); Metal's refit of a primitive acceleration structure is what this stands for.
// The tree keeps its shape: every packet takes its triangle's new vertices (k_flatten's records, by the id the packet carries), then the levels are walked bottom-up —
// one thread per node: the boxes of its leaf children from their triangles' padded boxes (k_flatten's, the build's own leaves; a pre-split triangle's references all get
// the whole triangle's box), those of its internal children from the level below, the node's grid and the children's planes by k_wide_level's rules.
__global__ void k_refit_wide_packets(const float4 *__restrict__ tri_world, float4 *__restrict__ wpackets, uint32_t n) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t gid = __float_as_uint(wpackets[WPK * (size_t)p].w);
    for (int j = 0; j < 3; j++) wpackets[WPK * (size_t)p + j] = tri_world[3 * (size_t)gid + j];
}
// A leaf child none of whose triangles moved (its instance's mesh was not updated) keeps the box it has — decoded from its planes on the node's old grid: the box the BUILD gave
// that reference, clipped to its slab if the triangle was pre-split (walls and floor: 32 references each; with the whole triangle's box on every one of them the refitted
// DragonScene rendered 14 % slower than a fresh build at a deformation of half a percent of the dragon's size).
__global__ void k_refit_wide_level(float4 *__restrict__ wnodes, const float4 *__restrict__ wpackets, const float4 *__restrict__ tri_lo, const float4 *__restrict__ tri_hi,
                                   const uint4 *__restrict__ tri_shade, const uint8_t *__restrict__ inst_dirty, float4 *__restrict__ nbox, uint32_t first, uint32_t count, double *__restrict__ growth /* [0] += area of the moved leaf children's boxes as they were, [1] += as they are now */) {
    __shared__ double s_g[2];          // (one wave per workgroup) the workgroup's two sums: one pair of global atomics per 64 nodes
    if (threadIdx.x == 0) { s_g[0] = 0.0; s_g[1] = 0.0; }
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double g_old = 0.0, g_new = 0.0;
    const size_t w = WNODE_STRIDE * (size_t)(first + i);
    const float4 n0 = wnodes[w], n1 = wnodes[w + 1];
    const uint32_t imask = __float_as_uint(n0.w) >> 24, cbase = __float_as_uint(n1.x), tbase = __float_as_uint(n1.y), meta[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)};
    const float BIG = 3.0e38f;
    float clo[8][3], chi[8][3]; bool occ[8];
    float nl[3] = {BIG, BIG, BIG}, nh[3] = {-BIG, -BIG, -BIG};
    uint32_t rank = 0; bool any = false;
    for (int sl = 0; sl < 8; sl++) {
        float lo[3] = {BIG, BIG, BIG}, hi[3] = {-BIG, -BIG, -BIG};
        occ[sl] = false;
        if ((imask >> sl) & 1u) {
            const uint32_t c = cbase + rank++;
            const float4 a = nbox[2 * (size_t)c], b = nbox[2 * (size_t)c + 1];
            lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; hi[0] = b.x; hi[1] = b.y; hi[2] = b.z; occ[sl] = true;
        } else {
            const uint32_t m = (meta[sl >> 2] >> (8 * (sl & 3))) & 0xFFu, cnt = m >> 5, off = m & 31u;
            bool moved = false;
            for (uint32_t r = 0; r < cnt; r++) {
                const uint32_t gid = __float_as_uint(wpackets[WPK * (size_t)(tbase + off + r)].w);
                moved = moved || inst_dirty[tri_shade[gid].w >> 16] != 0;
                const float4 a = tri_lo[gid], b = tri_hi[gid];
                lo[0] = fminf(lo[0], a.x); lo[1] = fminf(lo[1], a.y); lo[2] = fminf(lo[2], a.z);
                hi[0] = fmaxf(hi[0], b.x); hi[1] = fmaxf(hi[1], b.y); hi[2] = fmaxf(hi[2], b.z);
                occ[sl] = true;
            }
            if (cnt != 0u) {          // the box this child has: planes q * 2^e + p on the node's grid as it stands (rounded outwards when they were written)
                const uint32_t ew = __float_as_uint(n0.w);
                const float org[3] = {n0.x, n0.y, n0.z};
                const float4 p2 = wnodes[w + 2], p3 = wnodes[w + 3], p4 = wnodes[w + 4];
                const uint32_t pl[6][2] = {{__float_as_uint(p2.x), __float_as_uint(p2.y)}, {__float_as_uint(p2.z), __float_as_uint(p2.w)}, {__float_as_uint(p3.x), __float_as_uint(p3.y)},
                                           {__float_as_uint(p3.z), __float_as_uint(p3.w)}, {__float_as_uint(p4.x), __float_as_uint(p4.y)}, {__float_as_uint(p4.z), __float_as_uint(p4.w)}};
                float had_lo[3], had_hi[3], now_lo[3], now_hi[3];          // the child's box as it was, and the new one rounded outwards onto the SAME (old) grid: like with like
                for (int a = 0; a < 3; a++) {
                    const float st = __builtin_ldexpf(1.0f, (int)(int8_t)((ew >> (8 * a)) & 0xFFu));
                    const float ql = (float)((pl[a][sl >> 2] >> (8 * (sl & 3))) & 0xFFu), qh = (float)((pl[3 + a][sl >> 2] >> (8 * (sl & 3))) & 0xFFu);
                    // (exact: a plane is p + q * 2^e with q < 256)  Never beyond its triangles' own boxes: a moving sibling changes the node's grid with every refit, and a box
                    // re-rounded outwards onto each new grid would creep; an unsplit triangle's leaf thus keeps exactly its box, a pre-split reference at worst ends at its triangle's
                    had_lo[a] = __builtin_fmaf(ql, st, org[a]); had_hi[a] = __builtin_fmaf(qh, st, org[a]);
                    now_lo[a] = __builtin_fmaf(floorf((lo[a] - org[a]) / st), st, org[a]); now_hi[a] = __builtin_fmaf(ceilf((hi[a] - org[a]) / st), st, org[a]);
                    if (!moved) { lo[a] = fmaxf(lo[a], had_lo[a]); hi[a] = fminf(hi[a], had_hi[a]); }
                }
                if (moved) {          // what the refit does to the moved meshes' leaves: their boxes' area before and after (MRTSceneStats.leaf_growth)
                    const float ox = fmaxf(had_hi[0] - had_lo[0], 0.0f), oy = fmaxf(had_hi[1] - had_lo[1], 0.0f), oz = fmaxf(had_hi[2] - had_lo[2], 0.0f);
                    const float nx_ = fmaxf(now_hi[0] - now_lo[0], 0.0f), ny_ = fmaxf(now_hi[1] - now_lo[1], 0.0f), nz_ = fmaxf(now_hi[2] - now_lo[2], 0.0f);
                    g_old += (double)(ox * oy + oy * oz + oz * ox); g_new += (double)(nx_ * ny_ + ny_ * nz_ + nz_ * nx_);
                }
            }
        }
        for (int a = 0; a < 3; a++) { clo[sl][a] = lo[a]; chi[sl][a] = hi[a]; if (occ[sl]) { nl[a] = fminf(nl[a], lo[a]); nh[a] = fmaxf(nh[a], hi[a]); } }
        any = any || occ[sl];
    }
    if (growth) {          // (the lanes of the wave are together here)
        if (g_new > 0.0) { atomicAdd(&s_g[0], g_old); atomicAdd(&s_g[1], g_new); }
        __syncthreads();
        if (threadIdx.x == 0 && s_g[1] > 0.0) { atomicAdd(&growth[0], s_g[0]); atomicAdd(&growth[1], s_g[1]); }
    }
    if (!any) { nbox[2 * (size_t)(first + i)] = make_float4(n0.x, n0.y, n0.z, 0.0f); nbox[2 * (size_t)(first + i) + 1] = make_float4(n0.x, n0.y, n0.z, 0.0f); return; }      // (a node without children: nothing to move)
    // the node's grid: p = lo, step 2^e >= extent / 255 per axis; a child's planes rounded outwards and checked against their decoded positions (as k_wide_level)
    uint32_t eb[3]; float inv_step[3], step[3];
    for (int a = 0; a < 3; a++) {
        const float sdiv = (nh[a] - nl[a]) / 255.0f;
        const uint32_t bits = __float_as_uint(sdiv);
        uint32_t e = (bits >> 23) + ((bits & 0x7FFFFFu) ? 1u : 0u);
        if (e < 1u) e = 1u; if (e > 254u) e = 254u;
        eb[a] = e; step[a] = __uint_as_float(e << 23); inv_step[a] = __uint_as_float((254u - e) << 23);
    }
    uint32_t q[6][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int sl = 0; sl < 8; sl++) {
        uint32_t ql[3] = {255, 255, 255}, qh[3] = {0, 0, 0};
        if (occ[sl]) for (int a = 0; a < 3; a++) {
            float fl = floorf((clo[sl][a] - nl[a]) * inv_step[a]), fh = ceilf((chi[sl][a] - nl[a]) * inv_step[a]);
            fl = fminf(fmaxf(fl, 0.0f), 255.0f); fh = fminf(fmaxf(fh, 0.0f), 255.0f);
            if (nl[a] + fl * step[a] > clo[sl][a] && fl > 0.0f) fl -= 1.0f;
            if (nl[a] + fh * step[a] < chi[sl][a] && fh < 255.0f) fh += 1.0f;
            ql[a] = (uint32_t)fl; qh[a] = (uint32_t)fh;
        }
        for (int a = 0; a < 3; a++) { q[a][sl >> 2] |= ql[a] << (8 * (sl & 3)); q[3 + a][sl >> 2] |= qh[a] << (8 * (sl & 3)); }
    }
    wnodes[w + 0] = make_float4(nl[0], nl[1], nl[2], __uint_as_float(((eb[0] - 127u) & 0xFFu) | (((eb[1] - 127u) & 0xFFu) << 8) | (((eb[2] - 127u) & 0xFFu) << 16) | (imask << 24)));
    wnodes[w + 2] = make_float4(__uint_as_float(q[0][0]), __uint_as_float(q[0][1]), __uint_as_float(q[1][0]), __uint_as_float(q[1][1]));
    wnodes[w + 3] = make_float4(__uint_as_float(q[2][0]), __uint_as_float(q[2][1]), __uint_as_float(q[3][0]), __uint_as_float(q[3][1]));
    wnodes[w + 4] = make_float4(__uint_as_float(q[4][0]), __uint_as_float(q[4][1]), __uint_as_float(q[5][0]), __uint_as_float(q[5][1]));
    nbox[2 * (size_t)(first + i)] = make_float4(nl[0], nl[1], nl[2], 0.0f); nbox[2 * (size_t)(first + i) + 1] = make_float4(nh[0], nh[1], nh[2], 0.0f);
}

static inline uint32_t cdiv(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

// ------------------------------------------------------------------ refit of a rope layout (the BLASes of a two-level scene keep one: the query API and the in-place fallbacks walk it)
// Same topology, new boxes: every internal node notes itself as its children's parent; then one thread per LEAF takes its box from its triangles' padded boxes (by the id each
// packet carries) and climbs — the second thread to arrive at a node (a counter per node) unions the children's boxes and goes on.  The hand-off is k_refit's: 16-byte
// write-through stores, drained before the agent-scope arrival, sc1 loads after it.  Escape links and near-child masks are the build's: order, not correctness.
__global__ void k_rope_refit(float4 *nodes, uint32_t n, const float4 *__restrict__ packets, const float4 *__restrict__ tri_lo, const float4 *__restrict__ tri_hi,
                             const uint32_t *__restrict__ parent, const uint2 *__restrict__ ab /* per node: its {a, b} words, copied before the pass */, uint32_t *__restrict__ arrived,
                             const uint4 *__restrict__ tri_shade, const uint8_t *__restrict__ inst_dirty /* both or neither: a leaf none of whose triangles' instances moved keeps the box it has (the clipped boxes of pre-split references survive) */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint2 w = ab[i];
    if (!(w.x & NODE_LEAF)) return;
    const __amdgpu_buffer_rsrc_t rn = refit_rsrc(nodes);
    const float BIG = 3.0e38f;
    float4 lo = make_float4(BIG, BIG, BIG, 0.0f), hi = make_float4(-BIG, -BIG, -BIG, 0.0f);
    bool moved = inst_dirty == nullptr;
    for (uint32_t r = 0; r < w.y; r++) {
        const uint32_t gid = __float_as_uint(packets[3 * (size_t)((w.x & 0x7FFFFFFFu) + r)].w);
        if (inst_dirty) moved = moved || inst_dirty[tri_shade[gid].w >> 16] != 0;
        const float4 l = tri_lo[gid], h = tri_hi[gid];
        lo.x = fminf(lo.x, l.x); lo.y = fminf(lo.y, l.y); lo.z = fminf(lo.z, l.z); hi.x = fmaxf(hi.x, h.x); hi.y = fmaxf(hi.y, h.y); hi.z = fmaxf(hi.z, h.z);
    }
    if (!moved) { lo = nodes[4 * (size_t)i]; hi = nodes[4 * (size_t)i + 1]; }          // (written by the build or an earlier refit, long before this launch)
    for (;;) {
        lo.w = __uint_as_float(w.x); hi.w = __uint_as_float(w.y);
        refit_st_wt(rn, 4u * i, lo); refit_st_wt(rn, 4u * i + 1u, hi);
        const uint32_t p = parent[i];
        if (p == NONE) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   // this node's stores have left the CU ...
        const uint32_t old = __hip_atomic_fetch_add(&arrived[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ... before the arrival is counted
        if (old == 0u) return;                       // the sibling subtree finishes this node
        asm volatile("" ::: "memory");
        w = ab[p];
        const uint32_t l = w.x, r = w.y & NODE_INDEX_MASK;
        const float4 llo = refit_ld_wt(rn, 4u * l), lhi = refit_ld_wt(rn, 4u * l + 1u), rlo = refit_ld_wt(rn, 4u * r), rhi = refit_ld_wt(rn, 4u * r + 1u);
        lo = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), 0.0f);
        hi = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.0f);
        i = p;
    }
}
// {a, b} of every rope node, and its children's parent links, in one pass
__global__ void k_rope_prepare(const float4 *__restrict__ nodes, uint32_t n, uint32_t *__restrict__ parent, uint2 *__restrict__ ab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = __float_as_uint(nodes[4 * (size_t)i].w), b = __float_as_uint(nodes[4 * (size_t)i + 1].w);
    ab[i] = make_uint2(a, b);
    if (i == 0) parent[0] = NONE;
    if (!(a & NODE_LEAF)) { parent[a] = i; parent[b & NODE_INDEX_MASK] = i; }
}

// SAH cost of the 8-wide tree AS IT LIES IN MEMORY — what a refit changes and the build's sah_cost (the binary tree's) cannot show: the sum over all child boxes, decoded from
// their planes as the traversal decodes them, of area x (c_node for an internal child: one more node visit; c_tri per triangle for a leaf child).  The root's own visit and the
// normalisation by the root's area are the host's (wide_tree_cost).  *sum = that sum; rbox[0 .. 5] = the box of node `root` (the union of its children's boxes), as order-preserving
// uints (f2ord) through atomicMin / atomicMax.
__global__ void k_wide_cost(const float4 *__restrict__ wnodes, uint32_t first, uint32_t count, uint32_t root, float c_node, float c_tri, double *__restrict__ sum, uint32_t *__restrict__ rbox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double mine = 0.0;
    if (i < count) {
        const size_t w = WNODE_STRIDE * (size_t)(first + i);
        const float4 n0 = wnodes[w], n1 = wnodes[w + 1], p2 = wnodes[w + 2], p3 = wnodes[w + 3], p4 = wnodes[w + 4];
        const uint32_t ew = __float_as_uint(n0.w), imask = ew >> 24, meta[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)};
        const float org[3] = {n0.x, n0.y, n0.z};
        const uint32_t pl[6][2] = {{__float_as_uint(p2.x), __float_as_uint(p2.y)}, {__float_as_uint(p2.z), __float_as_uint(p2.w)}, {__float_as_uint(p3.x), __float_as_uint(p3.y)},
                                   {__float_as_uint(p3.z), __float_as_uint(p3.w)}, {__float_as_uint(p4.x), __float_as_uint(p4.y)}, {__float_as_uint(p4.z), __float_as_uint(p4.w)}};
        float st[3];
        for (int a = 0; a < 3; a++) st[a] = __builtin_ldexpf(1.0f, (int)(int8_t)((ew >> (8 * a)) & 0xFFu));
        for (int sl = 0; sl < 8; sl++) {
            const uint32_t m = (meta[sl >> 2] >> (8 * (sl & 3))) & 0xFFu, cnt = m >> 5;
            const bool inner = ((imask >> sl) & 1u) != 0u;
            if (!inner && cnt == 0u) continue;
            float lo[3], hi[3];
            for (int a = 0; a < 3; a++) {
                lo[a] = __builtin_fmaf((float)((pl[a][sl >> 2] >> (8 * (sl & 3))) & 0xFFu), st[a], org[a]);
                hi[a] = __builtin_fmaf((float)((pl[3 + a][sl >> 2] >> (8 * (sl & 3))) & 0xFFu), st[a], org[a]);
            }
            const float dx = fmaxf(hi[0] - lo[0], 0.0f), dy = fmaxf(hi[1] - lo[1], 0.0f), dz = fmaxf(hi[2] - lo[2], 0.0f);
            const float area = 2.0f * (dx * dy + dy * dz + dz * dx);
            mine += (double)area * (inner ? (double)c_node : (double)c_tri * (double)cnt);
            if (first + i == root) for (int a = 0; a < 3; a++) { atomicMin(&rbox[a], f2ord(lo[a])); atomicMax(&rbox[3 + a], f2ord(hi[a])); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0 && mine != 0.0) atomicAdd(sum, mine);
}

// diagnostics: how full are the 8-wide nodes?  out[c] = nodes with c children (c = 0..8), out[9] = internal children, out[10] = leaf children, out[11] = triangles
__global__ void k_wide_histogram(const float4 *__restrict__ wnodes, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 n0 = wnodes[WNODE_STRIDE * (size_t)i], n1 = wnodes[WNODE_STRIDE * (size_t)i + 1];
    const uint32_t imask = __float_as_uint(n0.w) >> 24, meta[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)};
    uint32_t leaves = 0, tris = 0;
    for (int k = 0; k < 8; k++) { const uint32_t cnt = ((meta[k >> 2] >> (8 * (k & 3))) & 0xFFu) >> 5; if (cnt) { leaves++; tris += cnt; } }
    const uint32_t inner = (uint32_t)__popc(imask);
    atomicAdd(&out[inner + leaves], 1u); atomicAdd(&out[9], inner); atomicAdd(&out[10], leaves); atomicAdd(&out[11], tris);
}

}  // namespace

// cost of the subtree of 8-wide nodes [first, first + count) rooted at `root`, per unit of the root's area: (c_node x area(root) + k_wide_cost's sum) / area(root).  Blocks.
int wide_tree_cost(const float4 *wnodes, uint32_t first, uint32_t count, uint32_t root, float c_node, float c_tri, hipStream_t stream, float *out) {
    *out = 0.0f;
    if (count == 0) return MRT_OK;
    DevBuf<double> d_sum; DevBuf<uint32_t> d_box;
    MRT_HIP(d_sum.alloc(1)); MRT_HIP(d_box.alloc(6));
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    MRT_HIP(hipMemsetAsync(d_sum.p, 0, 8, stream));
    MRT_HIP(hipMemcpyAsync(d_box.p, init, sizeof init, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_wide_cost, dim3(cdiv(count, 256)), dim3(256), 0, stream, wnodes, first, count, root, c_node, c_tri, d_sum.p, d_box.p);
    double h_sum = 0.0; uint32_t h_box[6];
    MRT_HIP(hipMemcpyAsync(&h_sum, d_sum.p, 8, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipMemcpyAsync(h_box, d_box.p, sizeof h_box, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    MRT_HIP(hipGetLastError());
    float b[6];
    for (int k = 0; k < 6; k++) b[k] = ord2f(h_box[k]);
    const double dx = std::max(0.0f, b[3] - b[0]), dy = std::max(0.0f, b[4] - b[1]), dz = std::max(0.0f, b[5] - b[2]);
    const double area = 2.0 * (dx * dy + dy * dz + dz * dx);
    *out = area > 0.0 ? (float)((c_node * area + h_sum) / area) : 0.0f;
    return MRT_OK;
}

// Refit of ONE BLAS of a two-level scene in the scene's shared arrays (two_level.hip refit_two_level): the mesh's new object-space triangles (k_flatten under the identity),
// the BLAS's packets of both layouts rewritten by the id each carries, its 8-wide nodes [wnode_base, + wnodes) bottom-up level by level (k_refit_wide_level: child and packet
// indices in there are absolute, the triangle arrays are the BLAS's own), its rope nodes by k_rope_refit, its normals.  Leaves the BLAS's root box (object space) in root_lo / root_hi.
int refit_blas(const HostMesh &g, const BlasRange &br, hipStream_t stream, DeviceScene &out, float root_lo[3], float root_hi[3], float *ms_out, float *growth_out) {
    const size_t nv = g.positions.size() / 3, T = br.ntri;
    if (T == 0 || br.wnodes == 0 || g.normals.size() != g.positions.size()) { set_error("refit_blas: nothing to refit"); return MRT_ERR_STATE; }
    static_assert(WPK == 3, "k_refit_wide_packets serves both packet arrays at a stride of three float4");
    std::vector<SubRec> recs; std::vector<uint32_t> idx; idx.reserve(3 * T);
    size_t tb = 0;
    for (size_t s = 0; s < g.sub_indices.size(); s++) {
        const auto &ix = g.sub_indices[s];
        if (ix.empty()) continue;
        recs.push_back(SubRec{(uint32_t)tb, (uint32_t)(ix.size() / 3), (uint32_t)idx.size(), 0u, 0u, (uint32_t)s});
        idx.insert(idx.end(), ix.begin(), ix.end()); tb += ix.size() / 3;
    }
    if (tb != T) { set_error("refit_blas: the mesh's triangle count changed"); return MRT_ERR_STATE; }
    std::vector<float4> h_nrm(nv);
    for (size_t v = 0; v < nv; v++) h_nrm[v] = make_float4(g.normals[3 * v], g.normals[3 * v + 1], g.normals[3 * v + 2], 0.0f);
    const float4 ident[4] = {make_float4(1, 0, 0, 0), make_float4(0, 1, 0, 0), make_float4(0, 0, 1, 0), make_float4(0, 0, 0, 0)};
    ScratchArena arena; arena.chunk_bytes = ((size_t)T * (48 + 16 + 32 + 12) + nv * 12 + (size_t)out.wnodes.n / WNODE_STRIDE * 32 + (size_t)br.rope_nodes * 16 + ((size_t)1 << 20) + 255) & ~(size_t)255;
    DevBuf<float> d_pos; DevBuf<uint32_t> d_idx, d_recs, cbounds, parent, arrived; DevBuf<uint2> ab; DevBuf<float4> cols, tri_world, tri_lo, tri_hi, nbox; DevBuf<uint4> ts_tmp; DevBuf<uint8_t> dirty;
    MRT_HIP(d_pos.alloc_in(arena, 3 * nv)); MRT_HIP(d_idx.alloc_in(arena, idx.size())); MRT_HIP(d_recs.alloc_in(arena, 6 * recs.size())); MRT_HIP(cbounds.alloc_in(arena, 6)); MRT_HIP(cols.alloc_in(arena, 4));
    MRT_HIP(tri_world.alloc_in(arena, 3 * T)); MRT_HIP(tri_lo.alloc_in(arena, T)); MRT_HIP(tri_hi.alloc_in(arena, T)); MRT_HIP(ts_tmp.alloc_in(arena, T));
    MRT_HIP(nbox.alloc_in(arena, 2 * (out.wnodes.n / WNODE_STRIDE))); MRT_HIP(dirty.alloc_in(arena, 4));
    DevBuf<double> growth; MRT_HIP(growth.alloc_in(arena, 2)); MRT_HIP(hipMemsetAsync(growth.p, 0, 16, stream));
    MRT_HIP(parent.alloc_in(arena, std::max<size_t>(br.rope_nodes, 1))); MRT_HIP(arrived.alloc_in(arena, std::max<size_t>(br.rope_nodes, 1))); MRT_HIP(ab.alloc_in(arena, std::max<size_t>(br.rope_nodes, 1)));
    struct EventPair { hipEvent_t a = nullptr, b = nullptr; ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } evs;
    MRT_HIP(hipEventCreate(&evs.a)); MRT_HIP(hipEventCreate(&evs.b));
    MRT_HIP(hipMemcpyAsync(d_pos.p, g.positions.data(), 12 * nv, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(d_idx.p, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(d_recs.p, recs.data(), recs.size() * sizeof(SubRec), hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(cols.p, ident, sizeof ident, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemsetAsync(dirty.p, 1, 4, stream));                       // the one "instance" of the BLAS's own triangle arrays moved
    MRT_HIP(hipMemsetAsync(cbounds.p, 0, 24, stream));
    MRT_HIP(hipMemsetAsync(arrived.p, 0, arrived.bytes(), stream));
    MRT_HIP(hipEventRecord(evs.a, stream));
    const uint32_t T32 = (uint32_t)T;
    hipLaunchKernelGGL(k_flatten, dim3(cdiv(T32, 1024)), dim3(1024), 0, stream, reinterpret_cast<const SubRec *>(d_recs.p), (int)recs.size(), d_pos.p, d_idx.p, cols.p, T32, tri_world.p, ts_tmp.p, tri_lo.p, tri_hi.p, cbounds.p);
    // 8-wide layout
    hipLaunchKernelGGL(k_refit_wide_packets, dim3(cdiv(T32, 256)), dim3(256), 0, stream, tri_world.p, out.wpackets.p + WPK * (size_t)br.packet_base, T32);
    std::vector<uint32_t> first(br.wide_levels.size(), br.wnode_base);
    for (size_t L = 1; L < br.wide_levels.size(); L++) first[L] = first[L - 1] + br.wide_levels[L - 1];
    for (size_t L = br.wide_levels.size(); L-- > 0;)
        hipLaunchKernelGGL(k_refit_wide_level, dim3(cdiv(br.wide_levels[L], 64)), dim3(64), 0, stream, out.wnodes.p, out.wpackets.p, tri_lo.p, tri_hi.p, ts_tmp.p, dirty.p, nbox.p, first[L], br.wide_levels[L], growth.p);
    // rope layout
    float4 *const rn = out.bnodes.p + 4 * (size_t)br.node_base, *const rp = out.bnodes.p + out.bpackets_offset + 3 * (size_t)br.packet_base;
    hipLaunchKernelGGL(k_refit_wide_packets, dim3(cdiv(T32, 256)), dim3(256), 0, stream, tri_world.p, rp, T32);
    if (br.rope_nodes) {
        hipLaunchKernelGGL(k_rope_prepare, dim3(cdiv(br.rope_nodes, 256)), dim3(256), 0, stream, (const float4 *)rn, br.rope_nodes, parent.p, ab.p);
        hipLaunchKernelGGL(k_rope_refit, dim3(cdiv(br.rope_nodes, 256)), dim3(256), 0, stream, rn, br.rope_nodes, (const float4 *)rp, tri_lo.p, tri_hi.p, (const uint32_t *)parent.p, (const uint2 *)ab.p, arrived.p, (const uint4 *)nullptr, (const uint8_t *)nullptr);
    }
    MRT_HIP(hipEventRecord(evs.b, stream));
    MRT_HIP(hipMemcpyAsync(out.normals.p + br.vbase, h_nrm.data(), nv * 16, hipMemcpyHostToDevice, stream));
    float4 h_box[2];
    MRT_HIP(hipMemcpyAsync(h_box, nbox.p + 2 * (size_t)br.wnode_base, sizeof h_box, hipMemcpyDeviceToHost, stream));
    double h_growth[2] = {0.0, 0.0};
    MRT_HIP(hipMemcpyAsync(h_growth, growth.p, sizeof h_growth, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    if (growth_out) *growth_out = h_growth[0] > 0.0 ? (float)(h_growth[1] / h_growth[0]) : 1.0f;
    MRT_HIP(hipGetLastError());
    float ms = 0; MRT_HIP(hipEventElapsedTime(&ms, evs.a, evs.b));
    if (ms_out) *ms_out = ms;
    root_lo[0] = h_box[0].x; root_lo[1] = h_box[0].y; root_lo[2] = h_box[0].z; root_hi[0] = h_box[1].x; root_hi[1] = h_box[1].y; root_hi[2] = h_box[1].z;
    return MRT_OK;
}

int wide_histogram(const DeviceScene &sc, hipStream_t stream, uint32_t out12[12]) {
    memset(out12, 0, 48);
    if (sc.num_wnodes == 0) return MRT_OK;
    DevBuf<uint32_t> d; MRT_HIP(d.alloc(12));
    MRT_HIP(hipMemsetAsync(d.p, 0, 48, stream));
    hipLaunchKernelGGL(k_wide_histogram, dim3(cdiv(sc.num_wnodes, 256)), dim3(256), 0, stream, sc.wnodes.p, sc.num_wnodes, d.p);
    MRT_HIP(hipMemcpyAsync(out12, d.p, 48, hipMemcpyDeviceToHost, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    return MRT_OK;
}

SceneView DeviceScene::view() const {
    SceneView v{};
    v.nodes = nodes.p; v.packets = nodes.p ? nodes.p + packets_offset : nullptr; v.tri_shade = tri_shade.p; v.normals = normals.p;
    v.base_color = base_color.p; v.materials = materials.p; v.inst_cols = inst_cols.p; v.geom_base = geom_base.p; v.lights = lights.p;
    v.wnodes = wnodes.p; v.wpackets = wpackets.p; v.num_wnodes = num_wnodes;
    v.num_wpackets = wpackets.p ? (uint32_t)(wpackets.n / WPK) : 0u; v.num_wtlas = wtlas_index.p ? (uint32_t)wtlas_index.n : 0u;
    v.inst = inst.p; v.tlas_index = tlas_index.p; v.wtlas_index = wtlas_index.p; v.tri_packet = (num_inst && num_wnodes) ? tri_packet.p : nullptr; v.inst_box = num_inst ? inst_box.p : nullptr; v.bnodes = bnodes.p; v.bpackets = bnodes.p ? bnodes.p + bpackets_offset : nullptr; v.num_inst = num_inst;
    v.num_nodes = rope_nodes; v.num_tris = num_packets;      // entries of `packets` (>= triangles when long triangles were pre-split into references)
    v.light_count = light_count; v.max_sub = stats.max_submeshes;
    return v;
}

int upload_lights(const MRTLight *lights, int count, hipStream_t stream, DeviceScene &out) {
    std::vector<LightDev> h((size_t)std::max(count, 1));
    memset(h.data(), 0, h.size() * sizeof(LightDev));
    for (int i = 0; i < count; i++) {
        const MRTLight &L = lights[i];
        LightDev &d = h[i];
        int32_t ty = L.type;
        float tyf; memcpy(&tyf, &ty, 4);
        d.position = make_float4(L.position.x, L.position.y, L.position.z, tyf);
        d.color = make_float4(L.color.x, L.color.y, L.color.z, 0);
        d.forward = make_float4(L.forward.x, L.forward.y, L.forward.z, 0);
        d.right = make_float4(L.right.x, L.right.y, L.right.z, 0);
        d.up = make_float4(L.up.x, L.up.y, L.up.z, 0);
        // normalize(direction) with the un-fused mrt-math order (Raytracing.metal:311,324); cos(coneAngle) (:314)
        float dx = L.direction.x, dy = L.direction.y, dz = L.direction.z;
        float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
        d.dirn = make_float4(dx * inv, dy * inv, dz * inv, cosf(L.coneAngle));
    }
    MRT_HIP(out.lights.alloc(h.size()));
    MRT_HIP(hipMemcpyAsync(out.lights.p, h.data(), h.size() * sizeof(LightDev), hipMemcpyHostToDevice, stream));
    MRT_HIP(hipStreamSynchronize(stream));
    out.light_count = count;
    return MRT_OK;
}

// What the rope layout can address (scene_device.h): a child index is 24 bits wide (k_emit_nodes packs the right child as
// near-mask << 24 | index, NODE_INDEX_MASK), and traverse.h reaches nodes and packets through ONE 32-bit byte offset from the
// node base: packet k lives at 64 * (2T - 1) + 48 * k.  `nodes` = surviving node count after the SAH collapse (0 = not known
// yet: only the triangle-count bound is checked).
int layout_limits(uint64_t triangles, uint64_t nodes) {
    if (triangles >= (1ull << 26)) { set_error("scene too large: the builder supports fewer than 2^26 triangles"); return MRT_ERR_UNSUPPORTED; }
    if (triangles > 0 && 64ull * (2 * triangles - 1) + 48ull * triangles > 0xFFFFFFFFull) {
        set_error("scene too large: nodes + packets exceed the 32-bit byte offset of the rope traversal (about 24.4 M triangles)"); return MRT_ERR_UNSUPPORTED;
    }
    if (nodes > (uint64_t)NODE_INDEX_MASK) { set_error("scene too large: the BVH has more nodes than the 24-bit child index of the rope layout can address"); return MRT_ERR_UNSUPPORTED; }
    return MRT_OK;
}

void pack_material(const MRTMaterial &m, float4 *out3) {
    out3[0] = make_float4(m.baseColor.x, m.baseColor.y, m.baseColor.z, m.dissolve);
    out3[1] = make_float4(m.specular.x, m.specular.y, m.specular.z, m.specularExponent);
    out3[2] = make_float4(m.emission.x, m.emission.y, m.emission.z, m.refractionIndex);
}

int build_scene(const std::vector<HostMesh> &meshes_in, const BuildOptions &opt, hipStream_t stream, DeviceScene &out, bool only_transforms_changed, bool only_vertices_changed) {
    out.validate = opt.validate != 0;
    if (opt.instancing) return build_two_level(meshes_in, opt, stream, out);
    out.num_inst = 0; out.inst.release(); out.tlas_index.release(); out.wtlas_index.release(); out.tri_packet.release(); out.inst_box.release(); out.tlas_wcap = 0; out.blas_wdepth = 0; out.bnodes.release(); out.h_inst.clear();
    // an instance (mrt_scene_add_instance) takes its geometry from its source mesh; flattening gives every instance its own world-space copy
    std::vector<MeshRef> refs;
    for (auto &m : meshes_in) refs.push_back(MeshRef{m.source >= 0 ? &meshes_in[(size_t)m.source] : &m, m.xf});
    if (int rc = build_flat(refs, opt, stream, out, nullptr, only_transforms_changed, only_vertices_changed && opt.refit)) return rc;
    const auto tv = std::chrono::steady_clock::now();
    const int rc = out.validate ? validate_layout(out, stream, false) : MRT_OK;
    out.commit_ms[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv).count();
    return rc;
}

// One world-space BVH over the given (geometry, transform) pairs: the whole flattened scene, or one BLAS (a single mesh under the identity).
// Copies of tens of megabytes on the host (the staging fill of a first commit): a few threads, each pulling tasks of <= 2 MB; serial when the work is small or no thread is to be had
static void run_tasks(std::vector<std::function<void()>> &tasks, size_t bytes) {
    const unsigned want = bytes < ((size_t)4 << 20) ? 1u : std::min(4u, std::max(1u, std::thread::hardware_concurrency()));
    std::atomic<size_t> next{0};
    auto work = [&] { for (size_t i; (i = next.fetch_add(1)) < tasks.size();) tasks[i](); };
    std::vector<std::thread> th;
    for (unsigned k = 1; k < want; k++) { try { th.emplace_back(work); } catch (...) { break; } }
    work();
    for (auto &t : th) t.join();
}

int build_flat(const std::vector<MeshRef> &refs, const BuildOptions &opt_in, hipStream_t stream, DeviceScene &out, PinnedBuf *stage, bool geometry_unchanged, bool refit) {
    BuildOptions opt = opt_in;
    // a wide node addresses the triangles of its leaf children with a 32-bit mask (8 children x 4): with the 8-wide layout the leaf limit is at most 4 — a larger max_leaf
    // applies to scenes built without it (scene option wide = 0).  Until round 6 such a scene silently lost the 8-wide layout and rendered at half the rate.
    if (opt.wide && opt.max_leaf > 4) opt.max_leaf = 4;
    struct MeshView { const std::vector<float> &positions, &normals; const float *xf; const std::vector<std::vector<uint32_t>> &sub_indices; const std::vector<MRTMaterial> &sub_materials; };
    std::vector<MeshView> meshes;
    for (auto &r : refs) meshes.push_back(MeshView{r.g->positions, r.g->normals, r.xf, r.g->sub_indices, r.g->sub_materials});
    // ---- host-side concatenation (one upload per array)
    size_t V = 0, T = 0, NI = 0; int max_sub = 1;
    for (auto &m : meshes) {
        V += m.positions.size() / 3;
        max_sub = std::max<int>(max_sub, (int)m.sub_indices.size());
        for (auto &s : m.sub_indices) { T += s.size() / 3; NI += s.size(); }
    }
    const size_t I = meshes.size();
    if (V >= 0xFFFFFFF0ull || I >= 65536 || max_sub >= 65536) { set_error("scene too large (limits: 2^32 - 16 vertices, 65535 instances / submeshes)"); return MRT_ERR_UNSUPPORTED; }
    if (T >= (1ull << 26)) { set_error("scene too large: the builder supports fewer than 2^26 triangles"); return MRT_ERR_UNSUPPORTED; }
    const auto tw0 = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    for (double &c : out.commit_ms) c = 0.0;
    // one pinned staging area for everything that goes up (positions, normals as float4, indices, the small tables): filled in ONE pass straight from the caller's
    // arrays (no zero-filled intermediate vectors), copied to the device by DMA from pinned pages.  DragonScene: 24 MB; the pageable path took 4 of the commit's 8.4 ms.
    const size_t slots = std::max<size_t>(I * max_sub, 1);
    size_t nrec = 0;
    for (auto &m : meshes) for (auto &sx : m.sub_indices) if (!sx.empty()) nrec++;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_pos = 0, o_nrm = o_pos + up(std::max<size_t>(V * 3, 3) * 4), o_idx = o_nrm + up(std::max<size_t>(V, 1) * 16), o_rec = o_idx + up(std::max<size_t>(NI, 3) * 4),
                 o_cols = o_rec + up(std::max<size_t>(nrec, 1) * sizeof(SubRec)), o_base = o_cols + up(std::max<size_t>(I * 4, 4) * 16), o_mat = o_base + up(slots * 16), o_gb = o_mat + up(3 * slots * 16), o_end = o_gb + up(slots * 4);
    PinnedBuf &stg = stage ? *stage : out.stage;
    const size_t n_pos = std::max<size_t>(V * 3, 3), n_nrm = std::max<size_t>(V, 1), n_idx = std::max<size_t>(NI, 3), n_cols = std::max<size_t>(I * 4, 4);
    // the geometry is on the device already (the previous commit of this scene left it there) and only transforms changed since: nothing of it is staged or uploaded again —
    // the commit of an animated flattened scene is the build itself
    const bool keep_geometry = geometry_unchanged && out.g_pos.p && out.g_pos.n == n_pos && out.g_idx.n == n_idx && out.g_recs.n == 6 * std::max<size_t>(nrec, 1) && out.normals.p && out.normals.n == n_nrm;
    MRT_HIP(stg.reserve(keep_geometry ? o_end - o_cols : o_end));
    uint8_t *const S = (uint8_t *)stg.p - (keep_geometry ? o_cols : 0);          // (keep_geometry: only the small tables are staged, at the start of the area; the geometry pointers below are never used)
    float *const h_pos = (float *)(S + o_pos); float4 *const h_nrm = (float4 *)(S + o_nrm); uint32_t *const h_idx = (uint32_t *)(S + o_idx); SubRec *const recs = (SubRec *)(S + o_rec);
    float4 *const h_cols = (float4 *)(S + o_cols), *const h_base = (float4 *)(S + o_base), *const h_mat = (float4 *)(S + o_mat); uint32_t *const h_gbase = (uint32_t *)(S + o_gb);
    memset(h_base, 0, slots * 16); memset(h_mat, 0, 3 * slots * 16); memset(h_gbase, 0, slots * 4);
    if (!keep_geometry) {
        if (V == 0) { h_pos[0] = h_pos[1] = h_pos[2] = 0.0f; h_nrm[0] = make_float4(0, 0, 0, 0); }
        if (NI == 0) { h_idx[0] = h_idx[1] = h_idx[2] = 0u; }
    }
    if (I == 0) for (int c = 0; c < 4; c++) h_cols[c] = make_float4(0, 0, 0, 0);
    // the big copies (positions, normals float3 -> float4, indices) as tasks of <= 2 MB for a few threads; the small tables here
    // what the build reads (positions, indices) is filled first, by a few threads; the normals — read by the renderer only — are filled by one more thread meanwhile and go up behind the
    // topology's kernels (DragonScene: 1.65 ms of staging before the first kernel became 0.45)
    std::vector<std::function<void()>> tasks, tasks_n; size_t task_bytes = 0;
    constexpr size_t CH = (size_t)1 << 19;        // elements per task (2 MB of floats)
    size_t vb = 0, tb = 0, ib = 0, nr = 0;
    for (size_t mi = 0; mi < I; mi++) {
        const MeshView &m = meshes[mi];
        size_t nv = m.positions.size() / 3;
        if (!keep_geometry) {
            const float *sp = m.positions.data(), *sn = m.normals.data(); float *dp = &h_pos[vb * 3]; float4 *dn = &h_nrm[vb];
            for (size_t a = 0; a < nv * 3; a += CH) { const size_t c = std::min(CH, nv * 3 - a); tasks.push_back([=] { memcpy(dp + a, sp + a, c * 4); }); }
            for (size_t a = 0; a < nv; a += CH / 4) { const size_t c = std::min(CH / 4, nv - a); tasks_n.push_back([=] { for (size_t v = a; v < a + c; v++) dn[v] = make_float4(sn[v * 3], sn[v * 3 + 1], sn[v * 3 + 2], 0.0f); }); }
            task_bytes += nv * 12;
        }
        for (int c = 0; c < 4; c++) h_cols[mi * 4 + c] = make_float4(m.xf[c * 4 + 0], m.xf[c * 4 + 1], m.xf[c * 4 + 2], 0.0f);
        for (size_t g = 0; g < m.sub_indices.size(); g++) {
            const auto &ix = m.sub_indices[g];
            h_base[mi * max_sub + g] = make_float4(m.sub_materials[g].baseColor.x, m.sub_materials[g].baseColor.y, m.sub_materials[g].baseColor.z, 0.0f);
            pack_material(m.sub_materials[g], &h_mat[3 * (mi * max_sub + g)]);
            h_gbase[mi * max_sub + g] = (uint32_t)tb;
            if (ix.empty()) continue;
            if (!keep_geometry) {
                const uint32_t *si = ix.data(); uint32_t *di = &h_idx[ib];
                for (size_t a = 0; a < ix.size(); a += CH) { const size_t c = std::min(CH, ix.size() - a); tasks.push_back([=] { memcpy(di + a, si + a, c * 4); }); }
                task_bytes += ix.size() * 4;
                recs[nr++] = SubRec{(uint32_t)tb, (uint32_t)(ix.size() / 3), (uint32_t)ib, (uint32_t)vb, (uint32_t)mi, (uint32_t)g};
            }
            tb += ix.size() / 3; ib += ix.size();
        }
        vb += nv;
    }
    struct Joiner { std::thread t; void join() { if (t.joinable()) t.join(); } ~Joiner() { join(); } } normals_fill;      // joined on every return path
    if (!tasks_n.empty()) {
        try { normals_fill.t = std::thread([&tasks_n] { for (auto &f : tasks_n) f(); }); }
        catch (...) { for (auto &f : tasks_n) f(); }
    }
    run_tasks(tasks, task_bytes);
    // a refit keeps the tree (and what the statistics say about it): same triangle count as the build that made the layouts — the 8-wide one, the rope one (scene option rope = 1, or a
    // scene built without the 8-wide layout), or both: whatever is resident is refitted
    const bool wide_there = opt.wide && out.num_wnodes != 0 && out.wnodes.p && out.wpackets.p && !out.wide_levels.empty(), rope_there = out.nodes.p != nullptr && out.rope_nodes != 0;
    const bool do_refit = refit && out.refit_triangles == T && T != 0 && (wide_there || rope_there) && (wide_there || !opt.wide);
    const MRTSceneStats stats_before = out.stats;
    if (!do_refit) { out.wide_levels.clear(); out.refit_triangles = 0; out.refits = 0; }
    out.stats = MRTSceneStats{};
    out.stats.triangles = T; out.stats.vertices = V; out.stats.instances = (int32_t)I; out.stats.max_submeshes = max_sub;
    out.stats.max_leaf_tris = opt.max_leaf;
    out.commit_ms[0] = since(tw0);
    const auto tw1 = std::chrono::steady_clock::now();

    if (!keep_geometry) MRT_HIP(out.normals.alloc(n_nrm));
    MRT_HIP(out.base_color.alloc(slots)); MRT_HIP(out.materials.alloc(3 * slots));
    MRT_HIP(hipMemcpyAsync(out.materials.p, h_mat, 3 * slots * 16, hipMemcpyHostToDevice, stream));
    MRT_HIP(out.geom_base.alloc(slots));
    MRT_HIP(out.inst_cols.alloc(n_cols));
    MRT_HIP(out.tri_shade.alloc(std::max<size_t>(T, 1)));
    MRT_HIP(hipMemcpyAsync(out.base_color.p, h_base, slots * 16, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(out.geom_base.p, h_gbase, slots * 4, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(out.inst_cols.p, h_cols, n_cols * 16, hipMemcpyHostToDevice, stream));

    // the normals go up once their staging is filled, behind the topology's kernels on the same stream.  (A second stream for them — the copy beside the kernels — was built and measured:
    // made per scene it cost 6 ms per commit, a new stream's first use sets up a hardware queue; made once per context it cost nothing here and 25 % of the renderer's one-frame latency,
    // 1.85 against 1.51 ms: every kernel of a lone frame ran slower with one more stream in use on the device.  profiles/r04_build_probe.txt)
    auto upload_normals = [&]() -> int {
        if (keep_geometry) return MRT_OK;
        normals_fill.join();
        MRT_HIP(hipMemcpyAsync(out.normals.p, h_nrm, n_nrm * 16, hipMemcpyHostToDevice, stream));
        return MRT_OK;
    };
    if (T == 0) {       // empty scene: every ray misses
        if (int rc = upload_normals()) return rc;
        MRT_HIP(out.nodes.alloc(8)); out.packets_offset = 4;
        MRT_HIP(hipStreamSynchronize(stream));
        out.stats.bvh_nodes = 0; out.stats.bvh_leaves = 0; out.num_packets = 0;
        out.rope_nodes = 0; out.num_wnodes = 0; out.wide_depth = 0; out.wnodes.release(); out.wpackets.release();
        out.stats.scene_bytes = 0;
        return MRT_OK;
    }

    const uint32_t T32 = (uint32_t)T;
    ScratchArena arena;                                          // before the buffers that borrow from it
    arena.chunk_bytes = ((size_t)T * (576 + 48 + (opt.wide && opt.max_leaf <= 4 ? 16 * WNODE_STRIDE : 0)) + ((size_t)4 << 20) + 255) & ~(size_t)255;     // what a build of T triangles takes (~510 B per triangle + the 8-wide nodes' scratch): one allocation, more if pre-splitting adds references
    DevBuf<float4> tri_world, tri_lo, tri_hi, ref_lo, ref_hi, node_lo, node_hi;
    DevBuf<uint32_t> cbounds, vals_a, vals_b, ghist, parent, left, right, flags, ntri, size, new_index, leaf_offset, stat, ref_tri;
    DevBuf<uint64_t> keys_a, keys_b;
    DevBuf<float> cost;
    DevBuf<uint8_t> collapsed, mask;
    // the build's geometry inputs stay with the scene (15 MB for DragonScene): the next commit of the same geometry under new transforms reads them where they are
    if (!keep_geometry) { MRT_HIP(out.g_pos.alloc(n_pos)); MRT_HIP(out.g_idx.alloc(n_idx)); MRT_HIP(out.g_recs.alloc(6 * std::max<size_t>(nrec, 1))); }
    float *const d_pos_p = out.g_pos.p; uint32_t *const d_idx_p = out.g_idx.p; SubRec *const d_recs_p = reinterpret_cast<SubRec *>(out.g_recs.p);
    static_assert(sizeof(SubRec) == 24, "SubRec is six 32-bit words");
    MRT_HIP(tri_world.alloc_in(arena, 3 * (size_t)T32)); MRT_HIP(tri_lo.alloc_in(arena, T32)); MRT_HIP(tri_hi.alloc_in(arena, T32));
    MRT_HIP(cbounds.alloc_in(arena, 6)); MRT_HIP(stat.alloc_in(arena, 4));

    struct EventPair {           // destroyed on every return path
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evs;
    MRT_HIP(hipEventCreate(&evs.a)); MRT_HIP(hipEventCreate(&evs.b));
    const hipEvent_t ev0 = evs.a, ev1 = evs.b;
    if (!keep_geometry) {
        MRT_HIP(hipMemcpyAsync(d_pos_p, h_pos, n_pos * 4, hipMemcpyHostToDevice, stream));
        MRT_HIP(hipMemcpyAsync(d_idx_p, h_idx, n_idx * 4, hipMemcpyHostToDevice, stream));
        if (nrec) MRT_HIP(hipMemcpyAsync(d_recs_p, recs, nrec * sizeof(SubRec), hipMemcpyHostToDevice, stream));
    }
    MRT_HIP(hipEventRecord(ev0, stream));
    {
        uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0};
        MRT_HIP(hipMemcpyAsync(cbounds.p, init, sizeof init, hipMemcpyHostToDevice, stream));
    }
    MRT_HIP(hipMemsetAsync(stat.p, 0xFF, stat.bytes(), stream));       // [0] depth and [1] leaves are cleared below; [2] = root stays NONE until k_assign finds it
    MRT_HIP(hipMemsetAsync(stat.p, 0, 8, stream));
    const int B = 256;
    out.commit_ms[1] = since(tw1);
    const auto tw2 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_flatten, dim3(cdiv(T32, 1024)), dim3(1024), 0, stream, d_recs_p, (int)nrec, d_pos_p, d_idx_p, out.inst_cols.p, T32,
                       tri_world.p, out.tri_shade.p, tri_lo.p, tri_hi.p, cbounds.p);
    if (do_refit) {
        DevBuf<float4> nbox; MRT_HIP(nbox.alloc_in(arena, 2 * (size_t)std::max<uint32_t>(out.num_wnodes, 1u)));
        DevBuf<double> growth; MRT_HIP(growth.alloc_in(arena, 2)); MRT_HIP(hipMemsetAsync(growth.p, 0, 16, stream));
        DevBuf<uint8_t> inst_dirty; MRT_HIP(inst_dirty.alloc_in(arena, std::max<size_t>(I, 1)));
        std::vector<uint8_t> h_dirty(std::max<size_t>(I, 1), 0);
        for (size_t mi = 0; mi < I; mi++) h_dirty[mi] = refs[mi].g->dirty ? 1 : 0;
        MRT_HIP(hipMemcpyAsync(inst_dirty.p, h_dirty.data(), h_dirty.size(), hipMemcpyHostToDevice, stream));
        if (wide_there) {
            hipLaunchKernelGGL(k_refit_wide_packets, dim3(cdiv(out.num_packets, B)), dim3(B), 0, stream, tri_world.p, out.wpackets.p, out.num_packets);
            std::vector<uint32_t> first(out.wide_levels.size(), 0u);
            for (size_t L = 1; L < out.wide_levels.size(); L++) first[L] = first[L - 1] + out.wide_levels[L - 1];
            for (size_t L = out.wide_levels.size(); L-- > 0;)
                hipLaunchKernelGGL(k_refit_wide_level, dim3(cdiv(out.wide_levels[L], 64)), dim3(64), 0, stream, out.wnodes.p, out.wpackets.p, tri_lo.p, tri_hi.p, out.tri_shade.p, inst_dirty.p, nbox.p, first[L], out.wide_levels[L], growth.p);
        }
        if (rope_there) {          // [r6] the rope layout beside it (rope = 1) or alone (wide = 0): packets by the id they carry, boxes by k_rope_refit
            DevBuf<uint32_t> parent, arrived; DevBuf<uint2> ab;
            MRT_HIP(parent.alloc_in(arena, out.rope_nodes)); MRT_HIP(arrived.alloc_in(arena, out.rope_nodes)); MRT_HIP(ab.alloc_in(arena, out.rope_nodes));
            MRT_HIP(hipMemsetAsync(arrived.p, 0, arrived.bytes(), stream));
            float4 *const rp = out.nodes.p + out.packets_offset;
            hipLaunchKernelGGL(k_refit_wide_packets, dim3(cdiv(out.num_packets, B)), dim3(B), 0, stream, tri_world.p, rp, out.num_packets);
            hipLaunchKernelGGL(k_rope_prepare, dim3(cdiv(out.rope_nodes, B)), dim3(B), 0, stream, (const float4 *)out.nodes.p, out.rope_nodes, parent.p, ab.p);
            hipLaunchKernelGGL(k_rope_refit, dim3(cdiv(out.rope_nodes, B)), dim3(B), 0, stream, out.nodes.p, out.rope_nodes, (const float4 *)rp, tri_lo.p, tri_hi.p, (const uint32_t *)parent.p, (const uint2 *)ab.p, arrived.p,
                               (const uint4 *)out.tri_shade.p, (const uint8_t *)inst_dirty.p);
        }
        MRT_HIP(hipEventRecord(ev1, stream));
        float4 h_box[2];
        if (wide_there) MRT_HIP(hipMemcpyAsync(h_box, nbox.p, sizeof h_box, hipMemcpyDeviceToHost, stream));
        else MRT_HIP(hipMemcpyAsync(h_box, out.nodes.p, sizeof h_box, hipMemcpyDeviceToHost, stream));          // (the rope root's box: node 0)
        double h_growth[2] = {0.0, 0.0};
        MRT_HIP(hipMemcpyAsync(h_growth, growth.p, sizeof h_growth, hipMemcpyDeviceToHost, stream));
        if (int rc = upload_normals()) return rc;
        MRT_HIP(hipStreamSynchronize(stream));
        MRT_HIP(hipGetLastError());
        float ms = 0; MRT_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        out.stats = stats_before;          // the tree's shape, and what was measured on it
        out.stats.build_ms = ms;
        // what the refit did to the tree: the 8-wide tree's cost as it lies now against the build's (MRTSceneStats.wide_cost / wide_cost_built); sah_cost — the build's binary-tree figure — scaled alike
        if (wide_there) { if (int rc = wide_tree_cost(out.wnodes.p, 0, out.num_wnodes, 0, opt.wide_cost_node, opt.wide_cost_tri, stream, &out.stats.wide_cost)) return rc; }
        if (out.stats.wide_cost_built > 0.0f) out.stats.sah_cost = out.sah_cost_built * (out.stats.wide_cost / out.stats.wide_cost_built);
        out.stats.refits = out.refits + 1;
        // the moved meshes' leaf boxes against what they were before this refit, chained over the refits since the build: the view-independent cost above hardly moves when a small,
        // finely tessellated mesh in a large room loosens (DragonScene, 2 % deformation: wide_cost x 1.014, rate x 0.85) — this does
        out.stats.leaf_growth = stats_before.leaf_growth * (h_growth[0] > 0.0 ? (float)(h_growth[1] / h_growth[0]) : 1.0f);
        out.root_lo[0] = h_box[0].x; out.root_lo[1] = h_box[0].y; out.root_lo[2] = h_box[0].z; out.root_hi[0] = h_box[1].x; out.root_hi[1] = h_box[1].y; out.root_hi[2] = h_box[1].z;
        out.commit_ms[2] = since(tw2);
        out.refits++;
        // scene option refit_max_cost_ratio: a tree that refits have loosened beyond that factor of its build-time cost is built again, here (the commit then costs a build)
        if (opt.refit_max_cost_ratio > 0.0f && ((out.stats.wide_cost_built > 0.0f && out.stats.wide_cost > opt.refit_max_cost_ratio * out.stats.wide_cost_built) || out.stats.leaf_growth > opt.refit_max_cost_ratio))
            return build_flat(refs, opt, stream, out, stage, false, false);
        return MRT_OK;
    }
    // ---- references: the build's leaves.  One per triangle, or several for a triangle much longer than the mean (k_split_emit)
    uint32_t n = T32;
    const float4 *leaf_lo_p = tri_lo.p, *leaf_hi_p = tri_hi.p;
    const uint32_t *ref_tri_p = nullptr;
    if (opt.presplit > 0.0f && T32 >= 64) {
        DevBuf<unsigned long long> esum; DevBuf<uint32_t> cnt, off;
        MRT_HIP(esum.alloc_in(arena, 1)); MRT_HIP(cnt.alloc_in(arena, (size_t)T32 + 1)); MRT_HIP(off.alloc_in(arena, (size_t)T32 + 1));
        MRT_HIP(hipMemsetAsync(esum.p, 0, 8, stream));
        MRT_HIP(hipMemsetAsync(cnt.p + T32, 0, 4, stream));
        hipLaunchKernelGGL(k_extent_sum, dim3(cdiv(T32, 1024)), dim3(1024), 0, stream, tri_lo.p, tri_hi.p, cbounds.p, T32, esum.p);
        hipLaunchKernelGGL(k_split_count, dim3(cdiv(T32, B)), dim3(B), 0, stream, tri_world.p, tri_lo.p, tri_hi.p, cbounds.p, T32, esum.p, opt.presplit, 32u, cnt.p);
        {   // exclusive scan of the counts: per-block scan, scan of the block sums, add; tot.p = number of references
            const uint32_t nb = cdiv(T32, 1024);
            DevBuf<uint32_t> bsum, tot; MRT_HIP(bsum.alloc_in(arena, nb + 1)); MRT_HIP(tot.alloc_in(arena, 1));
            hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(1024), 0, stream, cnt.p, off.p, bsum.p, T32, nullptr);
            hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, stream, bsum.p, nb, nullptr);
            hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, stream, off.p, bsum.p, T32, tot.p, cnt.p, nullptr);
            MRT_HIP(hipMemcpyAsync(off.p + T32, tot.p, 4, hipMemcpyDeviceToDevice, stream));
            MRT_HIP(hipStreamSynchronize(stream));
        }
        uint32_t total = 0;
        MRT_HIP(hipMemcpyAsync(&total, off.p + T32, 4, hipMemcpyDeviceToHost, stream));
        MRT_HIP(hipStreamSynchronize(stream));
        if (total > T32 && (uint64_t)total <= 2ull * T32) {           // (more than twice the triangles: the criterion is wrong for this mesh; build unsplit)
            n = total;
            MRT_HIP(ref_lo.alloc_in(arena, n)); MRT_HIP(ref_hi.alloc_in(arena, n)); MRT_HIP(ref_tri.alloc_in(arena, n));
            hipLaunchKernelGGL(k_split_emit, dim3(cdiv(T32, B)), dim3(B), 0, stream, tri_world.p, tri_lo.p, tri_hi.p, cnt.p, off.p, T32, ref_lo.p, ref_hi.p, ref_tri.p);
            MRT_HIP(hipStreamSynchronize(stream));                    // cnt / off die at scope exit
            leaf_lo_p = ref_lo.p; leaf_hi_p = ref_hi.p; ref_tri_p = ref_tri.p;
        }
    }
    out.num_packets = n;
    if (n >= (1u << 26)) { set_error("scene too large: the builder supports fewer than 2^26 references"); return MRT_ERR_UNSUPPORTED; }
    const uint32_t nnodes = 2 * n - 1;
    const uint32_t leaf_base = n - 1;
    MRT_HIP(node_lo.alloc_in(arena, nnodes)); MRT_HIP(node_hi.alloc_in(arena, nnodes));
    MRT_HIP(keys_a.alloc_in(arena, n)); MRT_HIP(keys_b.alloc_in(arena, n)); MRT_HIP(vals_a.alloc_in(arena, n)); MRT_HIP(vals_b.alloc_in(arena, n));
    const uint32_t sort_blocks = cdiv(n, SORT_TILE);
    MRT_HIP(ghist.alloc_in(arena, 256 * (size_t)sort_blocks));
    MRT_HIP(parent.alloc_in(arena, nnodes)); MRT_HIP(left.alloc_in(arena, n)); MRT_HIP(right.alloc_in(arena, n)); MRT_HIP(flags.alloc_in(arena, nnodes));
    MRT_HIP(ntri.alloc_in(arena, nnodes)); MRT_HIP(size.alloc_in(arena, nnodes)); MRT_HIP(cost.alloc_in(arena, nnodes)); MRT_HIP(collapsed.alloc_in(arena, nnodes)); MRT_HIP(mask.alloc_in(arena, nnodes));
    MRT_HIP(new_index.alloc_in(arena, nnodes)); MRT_HIP(leaf_offset.alloc_in(arena, nnodes));
    MRT_HIP(hipMemsetAsync(flags.p, 0, flags.bytes(), stream));
    hipLaunchKernelGGL(k_morton, dim3(cdiv(n, B)), dim3(B), 0, stream, leaf_lo_p, leaf_hi_p, cbounds.p, n, keys_a.p, vals_a.p);
    // 8 passes of 8 bits over 64-bit keys; after the 8 swaps the sorted data is back in the first pair
    auto radix_sort = [&](uint64_t *ka, uint64_t *kb, uint32_t *va, uint32_t *vb, uint32_t count, uint32_t *hist) {
        const uint32_t nb = cdiv(count, SORT_TILE);
        for (int pass = 0; pass < 8; pass++) {
            int shift = pass * 8;
            hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(SORT_THREADS), 0, stream, ka, count, shift, nb, hist);
            hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, stream, hist, 256 * nb, nullptr);
            hipLaunchKernelGGL(k_sort_scatter, dim3(nb), dim3(SORT_THREADS), 0, stream, ka, va, kb, vb, hist, count, shift, nb);
            std::swap(ka, kb); std::swap(va, vb);
        }
    };
    if (opt.builder != 2) radix_sort(keys_a.p, keys_b.p, vals_a.p, vals_b.p, n, ghist.p);
    uint64_t *kin = keys_a.p; uint32_t *vin = vals_a.p;
    TreeArrays t{node_lo.p, node_hi.p, parent.p, left.p, right.p, flags.p, cost.p, ntri.p, size.p, collapsed.p, mask.p};
    if (n == 1) {
        MRT_HIP(hipMemsetAsync(parent.p, 0xFF, 4, stream));
    } else if (opt.builder == 2) {
        // the quality yardstick: topology from the host's binned-SAH builder (bvh_host_sah.cpp); everything after it is the device pipeline
        std::vector<float4> h_lo(n), h_hi(n);
        MRT_HIP(hipMemcpyAsync(h_lo.data(), leaf_lo_p, (size_t)n * 16, hipMemcpyDeviceToHost, stream));
        MRT_HIP(hipMemcpyAsync(h_hi.data(), leaf_hi_p, (size_t)n * 16, hipMemcpyDeviceToHost, stream));
        MRT_HIP(hipStreamSynchronize(stream));
        std::vector<uint32_t> h_order, h_left, h_right, h_parent;
        host_sah_topology(h_lo.data(), h_hi.data(), n, h_order, h_left, h_right, h_parent);
        MRT_HIP(hipMemcpyAsync(vals_a.p, h_order.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
        MRT_HIP(hipMemcpyAsync(left.p, h_left.data(), (size_t)(n - 1) * 4, hipMemcpyHostToDevice, stream));
        MRT_HIP(hipMemcpyAsync(right.p, h_right.data(), (size_t)(n - 1) * 4, hipMemcpyHostToDevice, stream));
        MRT_HIP(hipMemcpyAsync(parent.p, h_parent.data(), (size_t)nnodes * 4, hipMemcpyHostToDevice, stream));
        MRT_HIP(hipStreamSynchronize(stream));          // the host vectors die at scope exit
    } else if (opt.builder == 0) {
        hipLaunchKernelGGL(k_karras, dim3(cdiv(n - 1, B)), dim3(B), 0, stream, kin, (int)n, left.p, right.p, parent.p);
    } else {
        // PLOC rounds; cluster arrays double as scratch
        DevBuf<uint32_t> cid, ncid, nn, keep, pos, bsum, counter;
        DevBuf<float4> clo, chi, nlo, nhi;
        MRT_HIP(cid.alloc_in(arena, n)); MRT_HIP(ncid.alloc_in(arena, n)); MRT_HIP(nn.alloc_in(arena, n)); MRT_HIP(keep.alloc_in(arena, n)); MRT_HIP(pos.alloc_in(arena, n));
        MRT_HIP(bsum.alloc_in(arena, cdiv(n, 1024) + 1)); MRT_HIP(counter.alloc_in(arena, 4));
        MRT_HIP(clo.alloc_in(arena, n)); MRT_HIP(chi.alloc_in(arena, n)); MRT_HIP(nlo.alloc_in(arena, n)); MRT_HIP(nhi.alloc_in(arena, n));
        MRT_HIP(hipMemsetAsync(counter.p, 0, 16, stream));
        hipLaunchKernelGGL(k_ploc_init, dim3(cdiv(n, B)), dim3(B), 0, stream, n, leaf_base, vin, leaf_lo_p, leaf_hi_p, cid.p, clo.p, chi.p);
        uint32_t m = n;
        int guard = 0, round = 0;
        MRT_HIP(hipMemsetD32Async((hipDeviceptr_t)(counter.p + 1), (int)n, 1, stream));       // the count round 0 reads; rounds alternate between counter[1] and counter[2]
        while (m > PLOC_TAIL) {
            if (++guard > 4096) { set_error("PLOC did not converge"); return MRT_ERR_HIP; }
            // a batch of rounds on grids sized for the last count the host has seen; every kernel takes the round's count from the device
            for (int k = 0; k < PLOC_ROUNDS_PER_READBACK; k++, round++) {
                const uint32_t *m_in = counter.p + 1 + (round & 1); uint32_t *m_out = counter.p + 1 + ((round + 1) & 1);
                hipLaunchKernelGGL(k_ploc_nn, dim3(cdiv(m, B)), dim3(B), 0, stream, m, opt.ploc_radius, clo.p, chi.p, nn.p, m_in);
                hipLaunchKernelGGL(k_ploc_merge, dim3(cdiv(m, 1024)), dim3(1024), 0, stream, m, nn.p, cid.p, clo.p, chi.p, keep.p, ncid.p, nlo.p, nhi.p,
                                   counter.p, left.p, right.p, parent.p, node_lo.p, node_hi.p, m_in);
                uint32_t nb = cdiv(m, 1024);
                hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(1024), 0, stream, keep.p, pos.p, bsum.p, m, m_in);
                hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, stream, bsum.p, nb, m_in);
                hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, stream, pos.p, bsum.p, m, m_out, keep.p, m_in);
                hipLaunchKernelGGL(k_ploc_compact, dim3(cdiv(m, B)), dim3(B), 0, stream, m, keep.p, pos.p, ncid.p, nlo.p, nhi.p, cid.p, clo.p, chi.p, m_in);
            }
            uint32_t new_m = 0;
            MRT_HIP(hipMemcpyAsync(&new_m, counter.p + 1 + (round & 1), 4, hipMemcpyDeviceToHost, stream));
            MRT_HIP(hipStreamSynchronize(stream));
            if (new_m >= m || new_m == 0) { set_error("PLOC made no progress"); return MRT_ERR_HIP; }
            m = new_m;
        }
        hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(1024), 0, stream, m, opt.ploc_radius, cid.p, clo.p, chi.p, counter.p, left.p, right.p, parent.p, node_lo.p, node_hi.p);
        uint32_t made = 0;
        MRT_HIP(hipMemcpyAsync(&made, counter.p, 4, hipMemcpyDeviceToHost, stream));
        MRT_HIP(hipStreamSynchronize(stream));   // scratch buffers die at scope exit
        if (made == NONE) { set_error("PLOC made no progress"); return MRT_ERR_HIP; }
    }
    DevBuf<float> dpC; DevBuf<uint8_t> dpD;
    WideDP dp{nullptr, nullptr};
    if (opt.wide && opt.max_leaf <= 4 && opt.wide_collapse) {
        MRT_HIP(dpC.alloc_in(arena, 8 * (size_t)nnodes)); MRT_HIP(dpD.alloc_in(arena, 8 * (size_t)nnodes));
        dp.C = dpC.p; dp.dec = dpD.p;
    }
    DevBuf<float4> refit_aux;
    MRT_HIP(refit_aux.alloc_in(arena, nnodes));
    if (opt.refit_fenced) hipLaunchKernelGGL(k_refit<true>, dim3(cdiv(n, B)), dim3(B), 0, stream, t, vin, leaf_lo_p, leaf_hi_p, n, leaf_base, opt.max_leaf, opt.cost_trav, opt.cost_isect, refit_aux.p,
                                             dp, std::min(opt.max_leaf, 4), opt.wide_cost_node, opt.wide_cost_tri);
    else hipLaunchKernelGGL(k_refit<false>, dim3(cdiv(n, B)), dim3(B), 0, stream, t, vin, leaf_lo_p, leaf_hi_p, n, leaf_base, opt.max_leaf, opt.cost_trav, opt.cost_isect, refit_aux.p,
                            dp, std::min(opt.max_leaf, 4), opt.wide_cost_node, opt.wide_cost_tri);
    hipLaunchKernelGGL(k_assign, dim3(cdiv(nnodes, 1024)), dim3(1024), 0, stream, t, nnodes, new_index.p, leaf_offset.p, stat.p);
    // the triangle packets in the leaf order of the binary tree (scratch): the source of the 8-wide layout's packets, and of the rope layout's when that one is emitted too
    DevBuf<float4> pk_tmp; MRT_HIP(pk_tmp.alloc_in(arena, 3 * (size_t)n));
    float4 *const packets_p = pk_tmp.p;
    hipLaunchKernelGGL(k_emit_packets, dim3(cdiv(n, B)), dim3(B), 0, stream, vin, leaf_offset.p, leaf_base, n, tri_world.p, ref_tri_p, packets_p);
    // ---- stats: the root (k_assign found it), its size, cost and box, depth and leaf count in one small read-back
    DevBuf<uint32_t> summary; MRT_HIP(summary.alloc_in(arena, 12));
    hipLaunchKernelGGL(k_build_summary, dim3(1), dim3(1), 0, stream, t, stat.p, summary.p);
    MRT_HIP(hipEventRecord(ev1, stream));
    uint32_t h_sum[12];
    MRT_HIP(hipMemcpyAsync(h_sum, summary.p, sizeof h_sum, hipMemcpyDeviceToHost, stream));
    if (int rc = upload_normals()) return rc;          // (every kernel of the topology is enqueued by now: the fill has had that long)
    MRT_HIP(hipStreamSynchronize(stream));
    MRT_HIP(hipGetLastError());
    float ms = 0; MRT_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    const uint32_t root = h_sum[0];
    if (root >= nnodes) { set_error("BVH build produced no root"); return MRT_ERR_HIP; }
    const uint32_t h_size = h_sum[1]; const uint32_t h_stat[2] = {h_sum[3], h_sum[4]};
    float h_cost; memcpy(&h_cost, &h_sum[2], 4);
    float4 rlo, rhi; memcpy(&rlo.x, &h_sum[6], 12); memcpy(&rhi.x, &h_sum[9], 12);
    float dx = rhi.x - rlo.x, dy = rhi.y - rlo.y, dz = rhi.z - rlo.z;
    float area = 2.0f * (dx * dy + dy * dz + dz * dx);
    out.root_lo[0] = rlo.x; out.root_lo[1] = rlo.y; out.root_lo[2] = rlo.z; out.root_hi[0] = rhi.x; out.root_hi[1] = rhi.y; out.root_hi[2] = rhi.z;
    out.stats.bvh_nodes = h_size; out.rope_nodes = 0;
    out.stats.bvh_leaves = h_stat[1];
    out.stats.max_depth = (int32_t)h_stat[0];
    out.stats.sah_cost = area > 0 ? h_cost / area : 0.0f; out.sah_cost_built = out.stats.sah_cost;
    out.stats.build_ms = ms;
    out.stats.scene_bytes = (uint64_t)T * 16 + (uint64_t)V * 16 + (uint64_t)I * max_sub * 20 + (uint64_t)I * 64;
    out.num_wnodes = 0; out.wide_depth = 0;
    out.commit_ms[2] = since(tw2);
    const auto tw3 = std::chrono::steady_clock::now();
    // a wide node addresses its leaf triangles with a 32-bit mask: 8 leaf children x max_leaf triangles must fit
    if (opt.wide && opt.max_leaf <= 4) {
        // ---- 8-wide compressed layout, level by level (BFS numbering)
        // greedy: every wide node is an inner node of the collapsed binary tree and swallows at least one more; optimal: every wide node has
        // at least two children and every leaf child at least one triangle, so there are fewer wide nodes than triangles.  The array is trimmed below.
        const size_t max_w = opt.wide_collapse ? (size_t)n + 2 : (size_t)h_size / 2 + 2;
        DevBuf<uint32_t> fa, fb, lv; DevBuf<float4> wtmp;
        MRT_HIP(fa.alloc_in(arena, max_w)); MRT_HIP(fb.alloc_in(arena, max_w)); MRT_HIP(lv.alloc_in(arena, WIDE_LV_WORDS));
        MRT_HIP(wtmp.alloc_in(arena, WNODE_STRIDE * max_w));          // built in scratch (worst case: one node per reference), copied into an array of the size the tree has
        MRT_HIP(out.wpackets.alloc(WPK * (size_t)n));
        MRT_HIP(hipEventRecord(ev0, stream));
        MRT_HIP(hipMemsetAsync(lv.p, 0, lv.bytes(), stream));
        { const uint32_t one = 1; MRT_HIP(hipMemcpyAsync(lv.p, &one, 4, hipMemcpyHostToDevice, stream)); }
        MRT_HIP(hipMemcpyAsync(fa.p, &root, 4, hipMemcpyHostToDevice, stream));
        std::vector<uint32_t> h_lv(WIDE_LV_WORDS, 0u);
        uint32_t total = 0; int depth = 0;
        constexpr int LEVELS_PER_READBACK = 16;
        for (uint32_t L0 = 0; ; L0 += LEVELS_PER_READBACK) {
            if (L0 + LEVELS_PER_READBACK >= WIDE_LV_MAX) { set_error("wide BVH deeper than 4096 levels"); return MRT_ERR_UNSUPPORTED; }
            for (uint32_t L = L0; L < L0 + LEVELS_PER_READBACK; L++) {
                const size_t ub = L >= 8 ? max_w : std::min<size_t>(max_w, (size_t)1 << (3 * L));      // a level holds at most 8^L nodes
                uint32_t *fin = (L & 1) ? fb.p : fa.p, *fo = (L & 1) ? fa.p : fb.p;
                if (opt.wide_collapse) hipLaunchKernelGGL(k_wide_level<true>, dim3(cdiv(ub, 64)), dim3(64), 0, stream, t, dp, leaf_offset.p, fin, L, (uint32_t)max_w, fo, lv.p, wtmp.p, packets_p, out.wpackets.p);
                else hipLaunchKernelGGL(k_wide_level<false>, dim3(cdiv(ub, 64)), dim3(64), 0, stream, t, dp, leaf_offset.p, fin, L, (uint32_t)max_w, fo, lv.p, wtmp.p, packets_p, out.wpackets.p);
            }
            MRT_HIP(hipMemcpyAsync(h_lv.data(), lv.p, WIDE_LV_WORDS * 4, hipMemcpyDeviceToHost, stream));
            MRT_HIP(hipStreamSynchronize(stream));
            if (h_lv[WIDE_LV_ERROR]) { set_error("wide BVH build overflow"); return MRT_ERR_HIP; }
            if (h_lv[L0 + LEVELS_PER_READBACK] == 0) break;           // the last level of the batch left nothing to do
        }
        for (uint32_t L = 0; L < WIDE_LV_MAX && h_lv[L] != 0; L++) { depth++; total += h_lv[L]; }
        if (total > max_w) { set_error("wide BVH build overflow"); return MRT_ERR_HIP; }
        if (h_lv[WIDE_LV_PACKETS] != n) { set_error("wide BVH build lost triangles"); return MRT_ERR_HIP; }
        MRT_HIP(out.wnodes.alloc(WNODE_STRIDE * (size_t)std::max(total, 1u)));
        MRT_HIP(hipMemcpyAsync(out.wnodes.p, wtmp.p, WNODE_STRIDE * (size_t)total * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        MRT_HIP(hipEventRecord(ev1, stream));
        MRT_HIP(hipStreamSynchronize(stream));
        MRT_HIP(hipGetLastError());
        float wms = 0; MRT_HIP(hipEventElapsedTime(&wms, ev0, ev1));
        out.stats.build_ms += wms;
        // A tree deeper than the traversal's LDS stack can be made (nested, growing triangles: the agglomerative builder merges one pair per round and the binary tree is a chain —
        // tests/test_deep_tree.py) would fall back to the rope kernels at half the rate.  The radix tree over the same Morton order (builder 0) is at most 63 key bits + the index
        // splits of equal keys deep whatever the geometry, and its 8-wide collapse a fraction of that: such a scene is built again with it and keeps the 8-wide layout
        // (from WIDE_DEPTH_REBUILD levels on: DragonScene has 13, a scene beyond 48 is a chain, and every level costs 320 B of LDS per wave).
        if (depth > WIDE_DEPTH_REBUILD && opt.builder == 1 && !refit) { BuildOptions o2 = opt; o2.builder = 0; return build_flat(refs, o2, stream, out, stage, geometry_unchanged, false); }
        out.wide_depth = depth;
        out.wide_levels.assign(h_lv.begin(), h_lv.begin() + depth); out.refit_triangles = T;
        if (depth <= WIDE_STACK_MAX && total < (1u << 24)) out.num_wnodes = total;
        if (int rc = wide_tree_cost(out.wnodes.p, 0, total, 0, opt.wide_cost_node, opt.wide_cost_tri, stream, &out.stats.wide_cost)) return rc;
        out.stats.wide_cost_built = out.stats.wide_cost; out.sah_cost_built = out.stats.sah_cost; out.stats.leaf_growth = 1.0f;       // deeper than any LDS stack the kernels are launched with (or child_base beyond its 24 stack bits): the rope backend, reported by MRTSceneStats::wide_layout = 0
        out.stats.scene_bytes += (uint64_t)total * 16 * WNODE_STRIDE + (uint64_t)n * 48;
        out.stats.bvh_nodes = out.num_wnodes ? total : h_size;
        out.stats.max_depth = out.num_wnodes ? depth : out.stats.max_depth;
        if (!out.num_wnodes) { out.wnodes.release(); out.wpackets.release(); out.stats.scene_bytes -= (uint64_t)total * 16 * WNODE_STRIDE + (uint64_t)n * 48; }      // not usable: the rope layout below is what this scene gets
    }
    // ---- the rope layout (64-byte binary nodes with escape links + a second copy of the packets): only for scenes without the 8-wide layout, or on request (scene option rope = 1:
    // the A/B switches that walk it, the BLASes of a two-level scene).  Every ray of the default pipeline walks the 8-wide layout, so DragonScene keeps 48 instead of 148 MB of BVH.
    out.commit_ms[3] = since(tw3);
    const auto tw4 = std::chrono::steady_clock::now();
    out.nodes.release(); out.packets_offset = 0;
    if (opt.rope || !out.num_wnodes) {
        if (int rc = layout_limits(n, h_size)) return rc;       // its 32-bit byte offsets and 24-bit child index bound the scene (about 24.4 M references)
        MRT_HIP(hipEventRecord(ev0, stream));
        // one allocation: [nodes | packets], so the traversal addresses both from one base
        MRT_HIP(out.nodes.alloc(4 * (size_t)h_size + 3 * (size_t)n));
        out.packets_offset = 4 * (size_t)h_size;
        hipLaunchKernelGGL(k_emit_nodes, dim3(cdiv(nnodes, B)), dim3(B), 0, stream, t, nnodes, new_index.p, leaf_offset.p, out.nodes.p);
        MRT_HIP(hipMemcpyAsync(out.nodes.p + out.packets_offset, packets_p, 3 * (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        MRT_HIP(hipEventRecord(ev1, stream));
        MRT_HIP(hipStreamSynchronize(stream));
        MRT_HIP(hipGetLastError());
        float rms = 0; MRT_HIP(hipEventElapsedTime(&rms, ev0, ev1));
        out.stats.build_ms += rms;
        out.rope_nodes = h_size; out.refit_triangles = T;
        out.stats.scene_bytes += (uint64_t)h_size * 64 + (uint64_t)n * 48;
    }
    out.commit_ms[4] = since(tw4);
    return MRT_OK;
}

}  // namespace mrt
