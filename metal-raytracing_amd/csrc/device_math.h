// device_math.h — fp32 helpers for the HIP kernels ("mrt-math v1").
//
// The translation unit is compiled with -ffp-contract=off: a*b+c stays two roundings unless
// __builtin_fmaf is spelled.  sqrt and '/' are correctly rounded (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).  The *shading* helpers follow the reference's
// expression order (Raytracing.metal:41-147); the triangle test and the vertex transform use the
// fused forms.  Box tests may use any formulation (they only have to be conservative).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MRT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
MRT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MRT_DEV f3 mk3(float4 a) { return mk3(a.x, a.y, a.z); }
MRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
MRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
MRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
MRT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
MRT_DEV f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
MRT_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
MRT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MRT_DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MRT_DEV float length3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }
MRT_DEV f3 normalize3(f3 a) { float inv = 1.0f / __builtin_sqrtf(dot3(a, a)); return a * inv; }
MRT_DEV float saturatef(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
MRT_DEV float fdot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
MRT_DEV f3 fcross(f3 a, f3 b) {
    return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}

// ---- Halton (Raytracing.metal:41-56).  Bases come from the prime table (:27-33), held in constant memory.
// Per base: the magic multipliers of the exact digit extraction below and the fp32 reciprocal, evaluated at compile time (the kernels used to
// derive them per call: three emulated integer divisions and a correctly rounded float division per Halton value, ~10 % of k_shade).
struct HaltonBase { uint32_t b, M, M2, M1; float invB; };
struct HaltonTable { HaltonBase e[100]; };
constexpr HaltonTable make_halton_table() {
    constexpr int primes[100] = {
        2,   3,   5,   7,   11,  13,  17,  19,  23,  29,  31,  37,  41,  43,  47,  53,  59,  61,  67,  71,
        73,  79,  83,  89,  97,  101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173,
        179, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281,
        283, 293, 307, 311, 313, 317, 331, 337, 347, 349, 353, 359, 367, 373, 379, 383, 389, 397, 401, 409,
        419, 421, 431, 433, 439, 443, 449, 457, 461, 463, 467, 479, 487, 491, 499, 503, 509, 521, 523, 541};
    HaltonTable t{};
    for (int k = 0; k < 100; k++) {
        const uint32_t b = (uint32_t)primes[k];
        t.e[k].b = b;
        t.e[k].M = 0xFFFFFFFFu / b + 1u;                       // mulhi(u, M) = u / b for u < 2^22
        t.e[k].M2 = 0xFFFFFFFFu / (b * b) + 1u;                // mulhi(u, M2) = u / b^2 for u < 2^22 (used for b <= 23)
        t.e[k].M1 = (1u << 20) / b + 1u;                       // (r2 * M1) >> 20 = r2 / b for r2 < b^2 <= 529
        t.e[k].invB = 1.0f / (float)primes[k];                 // IEEE division, as the kernels' correctly rounded '/'
    }
    return t;
}
static __constant__ HaltonTable c_halton = make_halton_table();

// Same float recurrence as the reference: f *= 1/b; r += f * (i % b); i /= b.  gfx950 has no integer divide, and
// this function is ~20 % of the VALU work of the primary-ray kernel and most of k_shade, so the integer part is
// done exactly but cheaply:
//   * base 2 (dimension 0): every partial sum of the recurrence is exactly representable for i < 2^24, so the
//     result equals the bit reversal of i scaled by 2^-32 — two instructions instead of 21 loop trips;
//   * other bases: q = mulhi(i, M) with M = floor(2^32 / b) + 1 is the exact quotient for i * b_err < 2^32, i.e.
//     for every i < 2^22 and b <= 541 (error term M*b - 2^32 <= b);
//   * i >= 2^22 (more than 3 M accumulated frames): the plain loop; i <= 0: 0, as the reference's loop gives.
// All three produce the same digits, hence the same floats, as the loop in the oracle.
MRT_DEV float halton_dev(int i, int d) {
    if (i <= 0) return 0.0f;       // the reference's `while (i > 0)` never runs (Raytracing.metal:46): seed offset + frame index wrapped past 2^31
    if (d == 0 && i < (1 << 24)) return (float)__brev((uint32_t)i) * 2.3283064365386963e-10f;   // exact: <= 24 significant bits
    const HaltonBase hb = c_halton.e[d];
    const int b = (int)hb.b;
    const float invB = hb.invB;
    float f = 1.0f, r = 0.0f;
    if (i >= (1 << 22)) {
        while (i > 0) { f = f * invB; r = r + f * (float)(i % b); i = i / b; }
        return r;
    }
    uint32_t u = (uint32_t)i;
    if (b <= 23) {
        // two digits per trip: b^2 <= 529 keeps mulhi(u, M2) the exact quotient u / b^2 for u < 2^22 (u * (M2*b^2 - 2^32) < 2^32);
        // the two-digit remainder r2 < 529 splits with full-rate 24-bit multiplies: d1 = (r2 * M1) >> 20 is exact because
        // r2 * (M1*b - 2^20) < 529 * 23 < 2^20.  An odd digit count ends with d1 = 0: r + f*0 == r.
        // (32-bit integer multiplies run at quarter rate on gfx950; per digit this loop costs ~36 cycles of VALU issue instead of ~60.)
        const uint32_t b2 = (uint32_t)(b * b);
        const uint32_t M2 = hb.M2, M1 = hb.M1;
        while (u > 0) {
            const uint32_t q = __umulhi(u, M2);
            const uint32_t r2 = u - __umul24(q, b2);
            const uint32_t d1 = __umul24(r2, M1) >> 20;
            const uint32_t d0 = r2 - __umul24(d1, (uint32_t)b);
            f = f * invB; r = r + f * (float)d0;
            f = f * invB; r = r + f * (float)d1;
            u = q;
        }
        return r;
    }
    const uint32_t M = hb.M;
    while (u > 0) {
        const uint32_t q = __umulhi(u, M);
        const uint32_t rem = u - __umul24(q, (uint32_t)b);  // q < 2^22, b < 2^10: the 24-bit multiply is exact (and full rate)
        f = f * invB;
        r = r + f * (float)rem;
        u = q;
    }
    return r;
}

// ---- sin/cos(2*pi*u): quadrant reduction in turns (exact) + Taylor polynomials with fmaf.
// Stands in for Metal's sincos(2*M_PI_F*u) (Raytracing.metal:79-82).
MRT_DEV void sincos_2pi_dev(float u, float &s, float &c) {
    float x = u * 4.0f;
    float qf = floorf(x + 0.5f);
    float r = x - qf;
    float th = r * 1.57079637f;
    float s2 = th * th;
    float sp = __builtin_fmaf(s2, 2.75573192e-6f, -1.98412698e-4f);
    sp = __builtin_fmaf(s2, sp, 8.33333333e-3f);
    sp = __builtin_fmaf(s2, sp, -1.66666667e-1f);
    sp = __builtin_fmaf(s2 * th, sp, th);
    float cp = __builtin_fmaf(s2, -2.75573192e-7f, 2.48015873e-5f);
    cp = __builtin_fmaf(s2, cp, -1.38888889e-3f);
    cp = __builtin_fmaf(s2, cp, 4.16666667e-2f);
    cp = __builtin_fmaf(s2, cp, -0.5f);
    cp = __builtin_fmaf(s2, cp, 1.0f);
    int q = ((int)qf) & 3;
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    s = (q == 2 || q == 3) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}

// Raytracing.metal:78-88
MRT_DEV f3 sample_cosine_hemisphere_dev(float ux, float uy) {
    float sin_phi, cos_phi;
    sincos_2pi_dev(ux, sin_phi, cos_phi);
    float cos_theta = __builtin_sqrtf(uy);
    float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
    return mk3(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi);
}

// Raytracing.metal:132-147
MRT_DEV f3 align_hemisphere_dev(f3 s, f3 n) {
    f3 right = normalize3(cross3(n, mk3(0.0072f, 1.0f, 0.0034f)));
    f3 forward = cross3(right, n);
    return (s.x * right + s.y * n) + s.z * forward;
}

// per-pixel Halton offset in [0, 2^20) (Renderer.swift:259 uses arc4random; ours is a counter hash)
MRT_DEV uint32_t seed_hash_dev(uint32_t seed, uint32_t idx) {
    uint32_t h = idx * 0x9E3779B1u + seed * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h & 0xFFFFFu;
}

// Ray / hit queue traffic is streamed: written once by one kernel, read once by the next, ~0.85 GB per frame through 32 MB of L2 whose job is
// to hold the BVH for the traversal's gathers.  MRT_STREAM_NT marks those accesses non-temporal (global_load / global_store ... nt).
#ifndef MRT_STREAM_NT
#define MRT_STREAM_NT 1     // measured: +4.5 % steady state, +3.7 % at 20 steps (same box, three alternations)
#endif
#ifndef MRT_STREAM_NT2
#define MRT_STREAM_NT2 0    // also: the per-pixel sample buffer (read-modify-write by the shadow rays), the accumulation targets, the seed table
#endif
typedef float mrt_vf4 __attribute__((ext_vector_type(4)));
MRT_DEV float4 qload(const float4 *p) {
#if MRT_STREAM_NT
    const mrt_vf4 v = __builtin_nontemporal_load(reinterpret_cast<const mrt_vf4 *>(p)); return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
MRT_DEV void qstore(float4 *p, float4 a) {
#if MRT_STREAM_NT
    const mrt_vf4 v = {a.x, a.y, a.z, a.w}; __builtin_nontemporal_store(v, reinterpret_cast<mrt_vf4 *>(p));
#else
    *p = a;
#endif
}

MRT_DEV float4 q2load(const float4 *p) {
#if MRT_STREAM_NT2
    const mrt_vf4 v = __builtin_nontemporal_load(reinterpret_cast<const mrt_vf4 *>(p)); return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
MRT_DEV void q2store(float4 *p, float4 a) {
#if MRT_STREAM_NT2
    const mrt_vf4 v = {a.x, a.y, a.z, a.w}; __builtin_nontemporal_store(v, reinterpret_cast<mrt_vf4 *>(p));
#else
    *p = a;
#endif
}
MRT_DEV uint32_t q2load(const uint32_t *p) {
#if MRT_STREAM_NT2
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

MRT_DEV float xorsign(float v, uint32_t sgn) { return __uint_as_float(__float_as_uint(v) ^ sgn); }
