// mrt_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT.
//
// A plain C++17 / fp32 restatement of the reference's hot path, used ONLY by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker (and as the timed,
// non-target CPU baseline).  Nothing under metal-raytracing_amd/ links, imports or calls this.
//
// What it restates (reference file:line):
//   halton                         Raytracing.metal:41-56, primes :27-33
//   interpolateVertexAttribute     Raytracing.metal:60-73
//   sampleCosineWeightedHemisphere Raytracing.metal:78-88
//   sampleAreaLight                Raytracing.metal:94-128
//   alignHemisphereWithNormal      Raytracing.metal:132-147
//   raytracingKernel               Raytracing.metal:156-405 (skipping the dead sort :178-197)
//   camera / lights                Scene.swift:18-67
//   TRS matrices                   Mesh.swift:21-24, Utilities.swift:104-166
//   uniforms / targets / seeds     Renderer.swift:216-229, :231-262
//   tonemap                        Shaders.metal:39-52
// The two closed-source pieces — Apple's `intersector.intersect` (Raytracing.metal:244,:367) and
// the acceleration-structure build (Utilities.swift:55-56,77) — are restated from their
// mathematical definition: nearest (resp. any) ray/triangle intersection over all instances,
// opaque, no culling.  A brute-force all-triangles loop is the ground truth; the oracle's own
// binned-SAH BVH is checked against it in tests/test_oracle_kat.py.
//
// PARITY PINNING: the reference holds no tests, golden vectors or images and cannot
// be built here (Swift + Metal + ModelIO).  The oracle is pinned by the known-answer values
// derivable from the kernel source (Halton values, hemisphere/basis identities,
// camera scalars, dragon TRS rows, the analytic floor radiance) — tests/test_oracle_kat.py.
// Against the Metal renderer's *image* parity is UNPINNED (no image, arc4random seeds).
//
// Arithmetic contract ("mrt-math v1") — shared by this file and, independently
// restated, by the HIP kernels, so that GPU and oracle agree bit-for-bit:
//   * IEEE fp32, round-to-nearest, denormals kept, no contraction (-ffp-contract=off) except
//     where fmaf() is written explicitly (the triangle test and the vertex transform);
//   * sqrt and divide correctly rounded; sin/cos of 2*pi*u by the polynomial below;
//   * closest hit = global minimum t, ties broken by the lowest global triangle id
//     (instance-major, then geometry, then primitive), so the answer is traversal-order free.

#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <memory>
#include <thread>
#include <atomic>
#include <algorithm>
#include <limits>
#include <string>
#include "../include/mrt_abi.h"

namespace {

struct V3 { float x, y, z; };
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 v3(const MRTFloat3 &a) { return V3{a.x, a.y, a.z}; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
// plain (un-fused) products: ((ax*bx) + (ay*by)) + (az*bz)
static inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline V3 cross3(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(V3 a) { return sqrtf(dot3(a, a)); }
static inline V3 normalize3(V3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return a * inv; }
static inline float saturate(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
// fused forms used by the triangle test and the vertex transform
static inline float fdot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline V3 fcross(V3 a, V3 b) {
    return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

// ---------------------------------------------------------------- Halton (Raytracing.metal:27-56)
static const short kPrimes[100] = {
    2,   3,   5,   7,   11,  13,  17,  19,  23,  29,  31,  37,  41,  43,  47,  53,  59,  61,  67,  71,
    73,  79,  83,  89,  97,  101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173,
    179, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281,
    283, 293, 307, 311, 313, 317, 331, 337, 347, 349, 353, 359, 367, 373, 379, 383, 389, 397, 401, 409,
    419, 421, 431, 433, 439, 443, 449, 457, 461, 463, 467, 479, 487, 491, 499, 503, 509, 521, 523, 541};

static float halton(int i, int d) {
    int b = kPrimes[d];
    float f = 1.0f;
    float invB = 1.0f / (float)b;
    float r = 0.0f;
    while (i > 0) {
        f = f * invB;
        r = r + f * (float)(i % b);
        i = i / b;
    }
    return r;
}

// ---------------------------------------------------------------- sin/cos(2*pi*u), mrt-math v1
// Stands in for Metal's sincos(2*M_PI_F*u) (Raytracing.metal:79-82).  Quadrant reduction in
// "turns" is exact; the polynomials are Taylor series on |theta| <= pi/4 evaluated with fmaf.
static void sincos_2pi(float u, float *s_out, float *c_out) {
    float x = u * 4.0f;
    float qf = floorf(x + 0.5f);
    float r = x - qf;                       // exact, in [-0.5, 0.5]
    float th = r * 1.57079637f;             // pi/2 in fp32 (0x3FC90FDB)
    float s2 = th * th;
    float sp = fmaf(s2, 2.75573192e-6f, -1.98412698e-4f);
    sp = fmaf(s2, sp, 8.33333333e-3f);
    sp = fmaf(s2, sp, -1.66666667e-1f);
    sp = fmaf(s2 * th, sp, th);             // th + th^3 * P(s2)
    float cp = fmaf(s2, -2.75573192e-7f, 2.48015873e-5f);
    cp = fmaf(s2, cp, -1.38888889e-3f);
    cp = fmaf(s2, cp, 4.16666667e-2f);
    cp = fmaf(s2, cp, -0.5f);
    cp = fmaf(s2, cp, 1.0f);
    int q = ((int)qf) & 3;
    float s, c;
    if (q == 0)      { s = sp;  c = cp; }
    else if (q == 1) { s = cp;  c = -sp; }
    else if (q == 2) { s = -sp; c = -cp; }
    else             { s = -cp; c = sp; }
    *s_out = s; *c_out = c;
}

// Raytracing.metal:78-88
static V3 sample_cosine_hemisphere(float ux, float uy) {
    float sin_phi, cos_phi;
    sincos_2pi(ux, &sin_phi, &cos_phi);
    float cos_theta = sqrtf(uy);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    return v3(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi);
}

// Raytracing.metal:132-147
static V3 align_hemisphere(V3 s, V3 n) {
    V3 up = n;
    V3 right = normalize3(cross3(n, v3(0.0072f, 1.0f, 0.0034f)));
    V3 forward = cross3(right, up);
    return (s.x * right + s.y * up) + s.z * forward;
}

// per-pixel seed: the reference uses arc4random() % 2^20 (Renderer.swift:259); ours is a
// counter hash so runs are reproducible.
static uint32_t seed_hash(uint32_t seed, uint32_t idx) {
    uint32_t h = idx * 0x9E3779B1u + seed * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h & 0xFFFFFu;
}

// ---------------------------------------------------------------- matrices (Utilities.swift:104-166)
struct M4 { float c[4][4]; };   // c[col][row], column-major like simd
static M4 m4_identity() { M4 m; memset(&m, 0, sizeof m); for (int i = 0; i < 4; i++) m.c[i][i] = 1; return m; }
static M4 m4_mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r.c[j][i] = ((a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1]) + a.c[2][i] * b.c[j][2]) + a.c[3][i] * b.c[j][3];
    return r;
}
static M4 m4_translate(V3 t) { M4 m = m4_identity(); m.c[3][0] = t.x; m.c[3][1] = t.y; m.c[3][2] = t.z; return m; }
static M4 m4_scale(float s) { M4 m = m4_identity(); m.c[0][0] = s; m.c[1][1] = s; m.c[2][2] = s; return m; }
static M4 m4_rotate_axis(float radians, V3 axis) {      // Utilities.swift:113-126
    axis = normalize3(axis);
    float ct = cosf(radians), st = sinf(radians), ci = 1 - ct;
    float x = axis.x, y = axis.y, z = axis.z;
    M4 m = m4_identity();
    m.c[0][0] = ct + x * x * ci;     m.c[0][1] = y * x * ci + z * st; m.c[0][2] = z * x * ci - y * st;
    m.c[1][0] = x * y * ci - z * st; m.c[1][1] = ct + y * y * ci;     m.c[1][2] = z * y * ci + x * st;
    m.c[2][0] = x * z * ci + y * st; m.c[2][1] = y * z * ci - x * st; m.c[2][2] = ct + z * z * ci;
    return m;
}
static M4 m4_rotate(V3 r) {                              // Utilities.swift:140-142: Rx*Ry*Rz
    return m4_mul(m4_mul(m4_rotate_axis(r.x, v3(1, 0, 0)), m4_rotate_axis(r.y, v3(0, 1, 0))), m4_rotate_axis(r.z, v3(0, 0, 1)));
}
static M4 m4_trs(V3 p, V3 r, float s) {                  // Mesh.swift:21-24
    return m4_mul(m4_mul(m4_translate(p), m4_rotate(r)), m4_scale(s));
}
// world = M * (p,1), fused: mrt-math v1
static V3 xform_point(const M4 &m, V3 p) {
    return v3(fmaf(m.c[2][0], p.z, fmaf(m.c[1][0], p.y, m.c[0][0] * p.x)) + m.c[3][0],
              fmaf(m.c[2][1], p.z, fmaf(m.c[1][1], p.y, m.c[0][1] * p.x)) + m.c[3][1],
              fmaf(m.c[2][2], p.z, fmaf(m.c[1][2], p.y, m.c[0][2] * p.x)) + m.c[3][2]);
}
// (M * (n,0)).xyz, plain: ((c0*x + c1*y) + c2*z)   (Raytracing.metal:267)
static V3 xform_dir(const M4 &m, V3 n) {
    return v3((m.c[0][0] * n.x + m.c[1][0] * n.y) + m.c[2][0] * n.z,
              (m.c[0][1] * n.x + m.c[1][1] * n.y) + m.c[2][1] * n.z,
              (m.c[0][2] * n.x + m.c[1][2] * n.y) + m.c[2][2] * n.z);
}

// ---------------------------------------------------------------- scene
struct Tri { V3 v0, e1, e2; };                 // world space
struct TriRef { uint32_t i0, i1, i2; int32_t inst, geom, prim; };   // shading lookup (resource table)
struct Hit { float t; float U, V, ad; uint32_t gid; };

struct Mesh {
    std::vector<V3> pos, nrm;
    M4 xf;
    std::vector<std::vector<uint32_t>> sub_idx;
    std::vector<MRTMaterial> sub_mat;
    uint32_t vbase = 0;
    int source = -1;            // >= 0: an instance of that mesh (its own arrays are empty)
};

struct BNode { float lo[3], hi[3]; uint32_t left, right; uint32_t first, count; };

struct Scene {
    std::vector<Mesh> meshes;
    std::vector<MRTLight> lights;
    // committed
    std::vector<Tri> tris;          // gid order
    std::vector<TriRef> refs;
    std::vector<V3> normals;        // concatenated object-space normals
    std::vector<MRTMaterial> mats;  // [inst*maxSub + geom]
    int maxSub = 0;
    std::vector<BNode> nodes;
    std::vector<uint32_t> order;    // leaf triangle order -> gid
    bool committed = false;
    // two-level mode (the reference's instance acceleration structure, Renderer.swift:193-213): one object-space scene per distinct
    // mesh, a world->object matrix and a first global triangle id per instance; `tris` / `nodes` above stay empty
    bool instancing = false;
    std::vector<std::unique_ptr<Scene>> blas;
    struct Inst { int blas; bool valid; float rows[3][4]; uint32_t gid_base, ntri; };
    std::vector<Inst> insts;
    uint64_t total_tris = 0;
};

static inline bool tri_test(const Tri &tr, V3 o, V3 d, float tmin, float tmax, float *t_out, float *U_out, float *V_out, float *ad_out) {
    V3 p = fcross(d, tr.e2);
    float det = fdot(tr.e1, p);
    if (!(det != 0.0f)) return false;                 // parallel (or NaN)
    float ad = fabsf(det);
    uint32_t sgn; memcpy(&sgn, &det, 4); sgn &= 0x80000000u;
    V3 tv = o - tr.v0;
    float U = fdot(tv, p); { uint32_t b; memcpy(&b, &U, 4); b ^= sgn; memcpy(&U, &b, 4); }
    if (!(U >= 0.0f && U <= ad)) return false;
    V3 q = fcross(tv, tr.e1);
    float V = fdot(d, q); { uint32_t b; memcpy(&b, &V, 4); b ^= sgn; memcpy(&V, &b, 4); }
    if (!(V >= 0.0f && U + V <= ad)) return false;
    float T = fdot(tr.e2, q); { uint32_t b; memcpy(&b, &T, 4); b ^= sgn; memcpy(&T, &b, 4); }
    float t = T / ad;
    if (!(t >= tmin && t <= tmax)) return false;
    *t_out = t; *U_out = U; *V_out = V; *ad_out = ad;
    return true;
}

static inline void hit_consider(Hit &h, const Tri &tr, uint32_t gid, V3 o, V3 d, float tmin, float tmax) {
    float t, U, V, ad;
    float lim = h.gid == 0xFFFFFFFFu ? tmax : h.t;
    if (!tri_test(tr, o, d, tmin, lim, &t, &U, &V, &ad)) return;
    if (h.gid == 0xFFFFFFFFu || t < h.t || (t == h.t && gid < h.gid)) { h.t = t; h.U = U; h.V = V; h.ad = ad; h.gid = gid; }
}

// conservative slab test (Ize 2013 style padding); boxes are additionally padded at build time
struct RayPre { V3 o, inv; };
static inline float safe_inv(float d) {
    float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    return 1.0f / a;
}
static inline bool box_test(const float lo[3], const float hi[3], const RayPre &r, float tmin, float tmax, float *tn_out) {
    float tn = tmin, tf = tmax;
    const float o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
    for (int a = 0; a < 3; a++) {
        float t0 = (lo[a] - o[a]) * inv[a], t1 = (hi[a] - o[a]) * inv[a];
        float mn = t0 < t1 ? t0 : t1, mx = t0 < t1 ? t1 : t0;
        mx = mx * 1.0000005f + 1e-30f;
        tn = mn > tn ? mn : tn; tf = mx < tf ? mx : tf;
    }
    *tn_out = tn;
    return tn <= tf;
}

static void tri_bounds(const Tri &t, float lo[3], float hi[3]) {
    V3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
    float xs[3] = {a.x, b.x, c.x}, ys[3] = {a.y, b.y, c.y}, zs[3] = {a.z, b.z, c.z};
    lo[0] = std::min({xs[0], xs[1], xs[2]}); hi[0] = std::max({xs[0], xs[1], xs[2]});
    lo[1] = std::min({ys[0], ys[1], ys[2]}); hi[1] = std::max({ys[0], ys[1], ys[2]});
    lo[2] = std::min({zs[0], zs[1], zs[2]}); hi[2] = std::max({zs[0], zs[1], zs[2]});
    for (int k = 0; k < 3; k++) {       // pad: absorbs fp32 error of the triangle test vs the slab test
        float m = std::max(fabsf(lo[k]), fabsf(hi[k]));
        float e = 1e-5f * m + 1e-6f;
        lo[k] -= e; hi[k] += e;
    }
}

struct Builder {
    Scene &s;
    std::vector<float> blo, bhi, cen;   // per tri
    explicit Builder(Scene &sc) : s(sc) {}
    uint32_t build(uint32_t first, uint32_t count, int depth) {
        uint32_t ni = (uint32_t)s.nodes.size();
        s.nodes.push_back(BNode{});
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
        for (uint32_t k = first; k < first + count; k++) {
            uint32_t g = s.order[k];
            for (int a = 0; a < 3; a++) {
                lo[a] = std::min(lo[a], blo[g * 3 + a]); hi[a] = std::max(hi[a], bhi[g * 3 + a]);
                clo[a] = std::min(clo[a], cen[g * 3 + a]); chi[a] = std::max(chi[a], cen[g * 3 + a]);
            }
        }
        BNode n{}; memcpy(n.lo, lo, 12); memcpy(n.hi, hi, 12);
        auto area = [](const float l[3], const float h[3]) {
            float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
            return 2.0f * (dx * dy + dy * dz + dz * dx);
        };
        bool leaf = count <= 2 || depth > 60;
        int best_axis = -1, best_split = -1; float best_cost = 1e30f;
        const int NB = 16;
        if (!leaf) {
            for (int a = 0; a < 3; a++) {
                float ext = chi[a] - clo[a];
                if (!(ext > 0)) continue;
                float blo_[NB][3], bhi_[NB][3]; int cnt[NB];
                for (int b = 0; b < NB; b++) { cnt[b] = 0; for (int k = 0; k < 3; k++) { blo_[b][k] = 1e30f; bhi_[b][k] = -1e30f; } }
                float sc = NB / ext;
                for (uint32_t k = first; k < first + count; k++) {
                    uint32_t g = s.order[k];
                    int b = std::min(NB - 1, std::max(0, (int)((cen[g * 3 + a] - clo[a]) * sc)));
                    cnt[b]++;
                    for (int q = 0; q < 3; q++) { blo_[b][q] = std::min(blo_[b][q], blo[g * 3 + q]); bhi_[b][q] = std::max(bhi_[b][q], bhi[g * 3 + q]); }
                }
                float ra[NB]; int rc[NB];
                { float l[3] = {1e30f, 1e30f, 1e30f}, h[3] = {-1e30f, -1e30f, -1e30f}; int c = 0;
                  for (int b = NB - 1; b > 0; b--) { for (int q = 0; q < 3; q++) { l[q] = std::min(l[q], blo_[b][q]); h[q] = std::max(h[q], bhi_[b][q]); } c += cnt[b]; ra[b] = c ? area(l, h) : 0; rc[b] = c; } }
                { float l[3] = {1e30f, 1e30f, 1e30f}, h[3] = {-1e30f, -1e30f, -1e30f}; int c = 0;
                  for (int b = 0; b < NB - 1; b++) { for (int q = 0; q < 3; q++) { l[q] = std::min(l[q], blo_[b][q]); h[q] = std::max(h[q], bhi_[b][q]); } c += cnt[b];
                      if (c == 0 || rc[b + 1] == 0) continue;
                      float cost = area(l, h) * c + ra[b + 1] * rc[b + 1];
                      if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; } } }
            }
            if (best_axis < 0) leaf = count <= 8;           // all centroids equal
            else if (count <= 4 && best_cost >= area(lo, hi) * count) leaf = true;
        }
        if (leaf) { n.first = first; n.count = count; n.left = n.right = 0; s.nodes[ni] = n; return ni; }
        uint32_t mid;
        if (best_axis < 0) mid = first + count / 2;
        else {
            float ext = chi[best_axis] - clo[best_axis]; float sc = NB / ext;
            auto it = std::partition(s.order.begin() + first, s.order.begin() + first + count, [&](uint32_t g) {
                int b = std::min(NB - 1, std::max(0, (int)((cen[g * 3 + best_axis] - clo[best_axis]) * sc)));
                return b <= best_split; });
            mid = (uint32_t)(it - s.order.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        uint32_t l = build(first, mid - first, depth + 1);
        uint32_t r = build(mid, first + count - mid, depth + 1);
        n.left = l; n.right = r; n.count = 0; n.first = 0;
        s.nodes[ni] = n;
        return ni;
    }
};

static bool invert_affine(const M4 &m, float rows[3][4]);
static void scene_commit(Scene &s) {
    s.tris.clear(); s.refs.clear(); s.normals.clear(); s.mats.clear(); s.nodes.clear(); s.order.clear(); s.blas.clear(); s.insts.clear();
    s.maxSub = 0;
    auto geom = [&](size_t mi) -> const Mesh & { return s.meshes[mi].source >= 0 ? s.meshes[(size_t)s.meshes[mi].source] : s.meshes[mi]; };
    for (size_t mi = 0; mi < s.meshes.size(); mi++) s.maxSub = std::max<int>(s.maxSub, (int)geom(mi).sub_idx.size());
    if (s.maxSub < 1) s.maxSub = 1;
    s.mats.assign(s.meshes.size() * (size_t)s.maxSub, MRTMaterial{});
    std::vector<int> blas_of_src(s.meshes.size(), -1);
    uint32_t vbase = 0;
    for (size_t mi = 0; mi < s.meshes.size(); mi++) {
        Mesh &m = s.meshes[mi];
        const Mesh &g = geom(mi);
        m.vbase = vbase;
        for (auto &n : g.nrm) s.normals.push_back(n);                       // the shading tables stay per instance in both modes
        std::vector<V3> wp;
        if (!s.instancing) { wp.resize(g.pos.size()); for (size_t i = 0; i < g.pos.size(); i++) wp[i] = xform_point(m.xf, g.pos[i]); }
        const uint32_t gid_base = (uint32_t)s.refs.size();
        for (size_t gi = 0; gi < g.sub_idx.size(); gi++) {
            s.mats[mi * s.maxSub + gi] = g.sub_mat[gi];
            const auto &ix = g.sub_idx[gi];
            for (size_t p = 0; p + 2 < ix.size(); p += 3) {
                if (!s.instancing) { Tri t; t.v0 = wp[ix[p]]; t.e1 = wp[ix[p + 1]] - t.v0; t.e2 = wp[ix[p + 2]] - t.v0; s.tris.push_back(t); }
                s.refs.push_back(TriRef{vbase + ix[p], vbase + ix[p + 1], vbase + ix[p + 2], (int32_t)mi, (int32_t)gi, (int32_t)(p / 3)});
            }
        }
        vbase += (uint32_t)g.pos.size();
        if (s.instancing) {
            const size_t src = m.source >= 0 ? (size_t)m.source : mi;
            if (blas_of_src[src] < 0) {                                      // object-space scene of this geometry: one mesh under the identity
                blas_of_src[src] = (int)s.blas.size();
                std::unique_ptr<Scene> b(new Scene());
                Mesh bm; bm.pos = g.pos; bm.nrm = g.nrm; bm.xf = m4_identity(); bm.sub_idx = g.sub_idx; bm.sub_mat = g.sub_mat;
                b->meshes.push_back(std::move(bm));
                scene_commit(*b);
                s.blas.push_back(std::move(b));
            }
            Scene::Inst in{}; in.blas = blas_of_src[src]; in.gid_base = gid_base; in.ntri = (uint32_t)s.refs.size() - gid_base;
            in.valid = invert_affine(m.xf, in.rows);
            s.insts.push_back(in);
        }
    }
    s.total_tris = s.refs.size();
    size_t T = s.tris.size();
    s.order.resize(T);
    for (size_t i = 0; i < T; i++) s.order[i] = (uint32_t)i;
    if (T) {
        Builder b(s);
        b.blo.resize(T * 3); b.bhi.resize(T * 3); b.cen.resize(T * 3);
        for (size_t i = 0; i < T; i++) {
            tri_bounds(s.tris[i], &b.blo[i * 3], &b.bhi[i * 3]);
            for (int a = 0; a < 3; a++) b.cen[i * 3 + a] = 0.5f * (b.blo[i * 3 + a] + b.bhi[i * 3 + a]);
        }
        s.nodes.reserve(T * 2);
        b.build(0, (uint32_t)T, 0);
    }
    s.committed = true;
}

static Hit closest_bvh(const Scene &s, V3 o, V3 d, float tmin, float tmax) {
    Hit h; h.t = tmax; h.U = h.V = 0; h.ad = 1; h.gid = 0xFFFFFFFFu;
    if (s.nodes.empty()) return h;
    RayPre rp{o, v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z))};
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const BNode &n = s.nodes[stack[--sp]];
        float tn; float lim = h.gid == 0xFFFFFFFFu ? tmax : h.t;
        if (!box_test(n.lo, n.hi, rp, tmin, lim, &tn)) continue;
        if (n.count) { for (uint32_t k = n.first; k < n.first + n.count; k++) hit_consider(h, s.tris[s.order[k]], s.order[k], o, d, tmin, tmax); }
        else { if (sp < 126) { stack[sp++] = n.right; stack[sp++] = n.left; } }
    }
    return h;
}
static bool any_bvh(const Scene &s, V3 o, V3 d, float tmin, float tmax) {
    if (s.nodes.empty()) return false;
    RayPre rp{o, v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z))};
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const BNode &n = s.nodes[stack[--sp]];
        float tn;
        if (!box_test(n.lo, n.hi, rp, tmin, tmax, &tn)) continue;
        if (n.count) { for (uint32_t k = n.first; k < n.first + n.count; k++) { float t, U, V, ad; if (tri_test(s.tris[s.order[k]], o, d, tmin, tmax, &t, &U, &V, &ad)) return true; } }
        else { if (sp < 126) { stack[sp++] = n.right; stack[sp++] = n.left; } }
    }
    return false;
}
static Hit closest_brute(const Scene &s, V3 o, V3 d, float tmin, float tmax) {
    Hit h; h.t = tmax; h.U = h.V = 0; h.ad = 1; h.gid = 0xFFFFFFFFu;
    for (size_t g = 0; g < s.tris.size(); g++) hit_consider(h, s.tris[g], (uint32_t)g, o, d, tmin, tmax);
    return h;
}
static bool any_brute(const Scene &s, V3 o, V3 d, float tmin, float tmax) {
    for (size_t g = 0; g < s.tris.size(); g++) { float t, U, V, ad; if (tri_test(s.tris[g], o, d, tmin, tmax, &t, &U, &V, &ad)) return true; }
    return false;
}

// ---------------------------------------------------------------- two-level (instanced) scenes
// The closed-source intersector walks an instance acceleration structure and takes the ray into each instance's object space; its
// arithmetic is unknown.  Here: rows of [A^-1 | -A^-1 t] evaluated in double in a fixed operation order and rounded to float once, the ray
// taken into object space with the fused mrt-math form and its direction NOT renormalised (so t stays the world distance), triangles
// tested in object space, closest hit = global minimum t with ties to the lowest global triangle id (instance-major numbering).
static bool invert_affine(const M4 &m, float rows[3][4]) {
    const float *xf = &m.c[0][0];
    const double a00 = xf[0], a10 = xf[1], a20 = xf[2], a01 = xf[4], a11 = xf[5], a21 = xf[6], a02 = xf[8], a12 = xf[9], a22 = xf[10];
    const double tx = xf[12], ty = xf[13], tz = xf[14];
    const double c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const double c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const double det = a00 * c00 + a01 * c10 + a02 * c20;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double i[3][3] = {{c00 / det, c01 / det, c02 / det}, {c10 / det, c11 / det, c12 / det}, {c20 / det, c21 / det, c22 / det}};
    for (int r = 0; r < 3; r++) {
        rows[r][0] = (float)i[r][0]; rows[r][1] = (float)i[r][1]; rows[r][2] = (float)i[r][2];
        rows[r][3] = (float)(-(i[r][0] * tx + i[r][1] * ty + i[r][2] * tz));
    }
    return true;
}
static V3 to_object_point(const float r[3][4], V3 p) {
    return v3(fmaf(r[0][2], p.z, fmaf(r[0][1], p.y, r[0][0] * p.x)) + r[0][3],
              fmaf(r[1][2], p.z, fmaf(r[1][1], p.y, r[1][0] * p.x)) + r[1][3],
              fmaf(r[2][2], p.z, fmaf(r[2][1], p.y, r[2][0] * p.x)) + r[2][3]);
}
static V3 to_object_dir(const float r[3][4], V3 v) {
    return v3(fmaf(r[0][2], v.z, fmaf(r[0][1], v.y, r[0][0] * v.x)),
              fmaf(r[1][2], v.z, fmaf(r[1][1], v.y, r[1][0] * v.x)),
              fmaf(r[2][2], v.z, fmaf(r[2][1], v.y, r[2][0] * v.x)));
}
static Hit closest_instanced(const Scene &s, V3 o, V3 d, float tmin, float tmax, bool brute) {
    Hit h; h.t = tmax; h.U = h.V = 0; h.ad = 1; h.gid = 0xFFFFFFFFu;
    for (const auto &in : s.insts) {
        if (!in.valid || in.ntri == 0) continue;
        const V3 oo = to_object_point(in.rows, o), dd = to_object_dir(in.rows, d);
        const float lim = h.gid == 0xFFFFFFFFu ? tmax : h.t;
        const Scene &b = *s.blas[in.blas];
        const Hit hb = brute ? closest_brute(b, oo, dd, tmin, lim) : closest_bvh(b, oo, dd, tmin, lim);
        if (hb.gid == 0xFFFFFFFFu) continue;
        // instances are visited in ascending order of global id: an equal t further on never wins the tie
        if (h.gid == 0xFFFFFFFFu || hb.t < h.t) { h = hb; h.gid = in.gid_base + hb.gid; }
    }
    return h;
}
static bool any_instanced(const Scene &s, V3 o, V3 d, float tmin, float tmax, bool brute) {
    for (const auto &in : s.insts) {
        if (!in.valid || in.ntri == 0) continue;
        const V3 oo = to_object_point(in.rows, o), dd = to_object_dir(in.rows, d);
        const Scene &b = *s.blas[in.blas];
        if (brute ? any_brute(b, oo, dd, tmin, tmax) : any_bvh(b, oo, dd, tmin, tmax)) return true;
    }
    return false;
}
static Hit scene_closest(const Scene &s, V3 o, V3 d, float tmin, float tmax, bool brute) {
    if (s.instancing) return closest_instanced(s, o, d, tmin, tmax, brute);
    return brute ? closest_brute(s, o, d, tmin, tmax) : closest_bvh(s, o, d, tmin, tmax);
}
static bool scene_any(const Scene &s, V3 o, V3 d, float tmin, float tmax, bool brute) {
    if (s.instancing) return any_instanced(s, o, d, tmin, tmax, brute);
    return brute ? any_brute(s, o, d, tmin, tmax) : any_bvh(s, o, d, tmin, tmax);
}

// ---------------------------------------------------------------- the kernel (Raytracing.metal:156-405)
struct Counters { uint64_t closest = 0, shadow = 0; };

struct StageDump {          // optional per-bounce dump for stage-level parity tests
    float *buf = nullptr;   // [pixel][bounce][16] floats
    int max_bounces = 0;
};

static V3 trace_pixel(const Scene &s, const MRTUniforms &u, uint32_t sample_index, uint32_t offset, int px, int py, int max_bounces,
                      bool brute, Counters &cnt, float *dump, bool materials = false) {
    const float INF = std::numeric_limits<float>::infinity();
    int idx = (int)(offset + sample_index);           // == u.frameIndex unless sample-sharded
    float r0 = halton(idx, 0), r1 = halton(idx, 1);                       // :202-203
    float pxf = (float)px + r0, pyf = (float)py + r1;                      // :204
    float uvx = pxf / (float)u.width, uvy = pyf / (float)u.height;         // :207
    uvx = uvx * 2.0f - 1.0f; uvy = uvy * 2.0f - 1.0f;                      // :208
    V3 cr = v3(u.camera.right), cu = v3(u.camera.up), cf = v3(u.camera.forward);
    V3 org = v3(u.camera.position);                                        // :214
    V3 dir = normalize3((uvx * cr + uvy * cu) + cf);                       // :216-218
    V3 color = v3(1, 1, 1), accumulated = v3(0, 0, 0);                     // :226-227
    for (int bounce = 0; bounce < max_bounces; bounce++) {                 // :237
        cnt.closest++;
        Hit h = scene_closest(s, org, dir, 0.0f, INF, brute);   // :244
        if (dump) { float *d = dump + bounce * 16; d[0] = org.x; d[1] = org.y; d[2] = org.z; d[3] = dir.x; d[4] = dir.y; d[5] = dir.z;
                    d[6] = h.gid == 0xFFFFFFFFu ? -1.0f : h.t; uint32_t g = h.gid; memcpy(&d[7], &g, 4); }
        if (h.gid == 0xFFFFFFFFu) break;                                   // :246-247
        const TriRef &tr = s.refs[h.gid];
        const M4 &xf = s.meshes[tr.inst].xf;                               // :249-258
        float bu = h.U / h.ad, bv = h.V / h.ad;
        V3 P = org + dir * h.t;                                            // :261
        float bw = 1.0f - bu - bv;                                         // :63-64
        V3 n_obj = (bu * s.normals[tr.i1] + bv * s.normals[tr.i2]) + bw * s.normals[tr.i0];   // :66-72
        V3 n = normalize3(xform_dir(xf, n_obj));                           // :267-268
        V3 surf = v3(s.mats[(size_t)tr.inst * s.maxSub + tr.geom].baseColor);   // :262-269
        if (materials) {
            // ---- materials extension (renderer option materials = 1).  NOT in raytracingKernel: the reference only carries the fields
            // (ShaderTypes.h:99-107, filled by SubMesh.swift:37-54) and lists "more advanced materials (starting with refraction)" as open
            // work (README.md:8).  Defined here, restated by k_shade<true>: emission is added along the path; one extra Halton dimension
            // (2 + 5 * max_bounces + bounce, beyond the reference's) picks dielectric refraction / reflection (dissolve < 1), a specular
            // lobe (GGX half vector with alpha^2 = 2 / (Ns + 2); only +, *, /, sqrt and the mrt-math sincos, so both sides agree bit for bit)
            // or the reference's diffuse path with its albedo divided by the probability of having been chosen.
            const MRTMaterial &m = s.mats[(size_t)tr.inst * s.maxSub + tr.geom];
            accumulated = accumulated + color * v3(m.emission);
            const float ul = halton(idx, 2 + 5 * max_bounces + bounce);
            const V3 spec = v3(m.specular);
            const float kd = std::max(surf.x, std::max(surf.y, surf.z)), ks = std::max(spec.x, std::max(spec.y, spec.z));
            const float trn = (m.dissolve > 0.0f && m.dissolve < 1.0f && m.refractionIndex > 0.0f) ? 1.0f - m.dissolve : 0.0f;
            if (ul < trn) {                                                 // dielectric interface, clear (throughput unchanged), no next-event estimate
                const float u2 = ul / trn;
                const float cd = dot3(dir, n);
                const bool entering = cd < 0.0f;
                const V3 nn = entering ? n : neg(n);
                const float ni = m.refractionIndex;
                const float eta = entering ? 1.0f / ni : ni;
                const float cosi = entering ? -cd : cd;
                const float sin2t = (eta * eta) * (1.0f - cosi * cosi);
                float r0 = (1.0f - ni) / (1.0f + ni); r0 = r0 * r0;
                float F = 1.0f, cost = 0.0f;
                if (sin2t < 1.0f) { cost = sqrtf(1.0f - sin2t); const float c = entering ? cosi : cost; const float x = 1.0f - c; const float x2 = x * x; F = r0 + (1.0f - r0) * ((x2 * x2) * x); }
                V3 nd;
                if (u2 < F) { nd = dir + nn * (2.0f * cosi); org = P + nn * 1e-3f; }
                else { nd = dir * eta + nn * (eta * cosi - cost); org = P + nn * -1e-3f; }
                dir = normalize3(nd);
                if (dump) { float *d = dump + bounce * 16; d[8] = n.x; d[9] = n.y; d[10] = n.z; d[11] = d[12] = d[13] = 0.0f; d[14] = -2.0f; d[15] = -1.0f; }
                continue;
            }
            const float ud = trn > 0.0f ? (ul - trn) / (1.0f - trn) : ul;
            const float ps = (ks > 0.0f && m.specularExponent > 0.0f) ? ks / (ks + kd) : 0.0f;
            if (ud < ps) {                                                  // specular lobe: reflect about a GGX-distributed half vector, no next-event estimate
                const float hx = halton(idx, 2 + bounce * 5 + 3), hy = halton(idx, 2 + bounce * 5 + 4);
                const float a2 = 2.0f / (m.specularExponent + 2.0f);
                const float ct2 = (1.0f - hy) / (1.0f + (a2 - 1.0f) * hy);
                const float ct = sqrtf(ct2), st = sqrtf(1.0f - ct2);
                float sp_, cp_; sincos_2pi(hx, &sp_, &cp_);
                const V3 hw = align_hemisphere(v3(st * cp_, ct, st * sp_), n);
                const float dh = dot3(dir, hw);
                const V3 wi = dir - hw * (2.0f * dh);
                if (dump) { float *d = dump + bounce * 16; d[8] = n.x; d[9] = n.y; d[10] = n.z; d[11] = d[12] = d[13] = 0.0f; d[14] = -3.0f; d[15] = -1.0f; }
                if (!(dot3(wi, n) > 0.0f)) break;                           // sampled below the surface: the path is absorbed
                color = color * (spec * (1.0f / ps));
                org = P + n * 1e-3f; dir = normalize3(wi);
                continue;
            }
            if (ps > 0.0f) surf = surf * (1.0f / (1.0f - ps));             // diffuse lobe chosen with probability 1 - ps
        }
        float ls = halton(idx, 2 + bounce * 5 + 0);                        // :272
        int li = std::min((int)(ls * (float)u.lightCount), u.lightCount - 1);   // :273
        const MRTLight &L = s.lights[li];
        V3 ldir, lcol; float ldist;
        if (L.type == MRTLightTypeAreaLight) {                             // :281-290, :94-128
            float ax = halton(idx, 2 + bounce * 5 + 1) * 2.0f - 1.0f;
            float ay = halton(idx, 2 + bounce * 5 + 2) * 2.0f - 1.0f;
            V3 sp = (v3(L.position) + v3(L.right) * ax) + v3(L.up) * ay;
            ldir = sp - P;
            ldist = length3(ldir);
            float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
            ldir = ldir * inv;
            lcol = v3(L.color) * (inv * inv);
            lcol = lcol * saturate(dot3(neg(ldir), v3(L.forward)));
        } else if (L.type == MRTLightTypeSpotlight) {                      // :292-316
            ldir = v3(L.position) - P;
            ldist = length3(ldir);
            float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
            ldir = ldir * inv;
            lcol = v3(0, 0, 0);
            V3 cone = normalize3(v3(L.direction));
            float spot = dot3(neg(ldir), cone);
            if (spot > cosf(L.coneAngle)) lcol = (v3(L.color) * inv) * inv;
        } else if (L.type == MRTLightTypePointlight) {                     // :317-322
            ldir = v3(L.position) - P;
            ldist = length3(ldir);
            float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
            ldir = ldir * inv;
            lcol = (v3(L.color) * inv) * inv;
        } else {                                                           // :323-327 sunlight
            ldir = neg(normalize3(v3(L.direction)));
            ldist = INF;
            lcol = v3(L.color);
        }
        lcol = lcol * saturate(dot3(n, ldir));                             // :331
        lcol = lcol * (float)u.lightCount;                                 // :335
        color = color * surf;                                              // :339
        int shadowed = -1;
        if (length3(lcol) > 0.0001f) {                                     // :341
            V3 so = P + n * 1e-3f;                                         // :350
            float smax = ldist - 1e-3f;                                    // :356
            cnt.shadow++;
            bool occ = scene_any(s, so, ldir, 0.0f, smax, brute);   // :367
            shadowed = occ ? 1 : 0;
            if (!occ) accumulated = accumulated + lcol * color;            // :371-373
        }
        if (dump) { float *d = dump + bounce * 16; d[8] = n.x; d[9] = n.y; d[10] = n.z; d[11] = lcol.x; d[12] = lcol.y; d[13] = lcol.z; d[14] = (float)shadowed; d[15] = (float)li; }
        float hx = halton(idx, 2 + bounce * 5 + 3), hy = halton(idx, 2 + bounce * 5 + 4);   // :384-385
        V3 sd = align_hemisphere(sample_cosine_hemisphere(hx, hy), n);     // :387-388
        org = P + n * 1e-3f;                                               // :390
        dir = sd;                                                          // :391
    }
    return accumulated;
}

struct Renderer {
    const Scene *scene;
    int w, h, max_bounces; uint32_t seed;
    MRTCamera cam;
    std::vector<uint32_t> seeds;
    std::vector<float> accum;      // RGBA32F
    uint32_t frameIndex = 0;
    Counters total;
    int shard_rank = 0, shard_world = 1;
    uint32_t sample_offset = 0;
    bool materials = false;        // the materials extension (see trace_pixel)
};

static void default_camera(int w, int h, MRTCamera *c) {       // Scene.swift:40-57
    memset(c, 0, sizeof *c);
    c->position = MRTFloat3{0.0f, 1.0f, 5.38f, 0};
    float fov = 45.0f * (3.14159274f / 180.0f);
    float aspect = (float)w / (float)h;
    float ih = tanf(fov / 2.0f);
    float iw = aspect * ih;
    c->right = MRTFloat3{1.0f * iw, 0.0f * iw, 0.0f * iw, 0};
    c->up = MRTFloat3{0.0f * ih, 1.0f * ih, 0.0f * ih, 0};
    c->forward = MRTFloat3{0.0f, 0.0f, -1.0f, 0};
}

static void render_frames(Renderer &r, int nframes, int nthreads, bool brute, float *dump) {
    const Scene &s = *r.scene;
    for (int f = 0; f < nframes; f++) {
        MRTUniforms u{}; u.width = r.w; u.height = r.h; u.blocksWide = (r.w + 15) / 16;     // Renderer.swift:216-229
        u.frameIndex = r.frameIndex; u.lightCount = (int)s.lights.size(); u.camera = r.cam;
        std::atomic<int> next_row{0};
        std::vector<Counters> cnts(nthreads);
        auto work = [&](int tid) {
            for (;;) {
                int y = next_row.fetch_add(1);
                if (y >= r.h) break;
                for (int x = 0; x < r.w; x++) {
                    if (r.shard_world > 1) {
                        int tile = (y / 8) * ((r.w + 7) / 8) + (x / 8);
                        if (tile % r.shard_world != r.shard_rank) continue;
                    }
                    size_t p = (size_t)y * r.w + x;
                    float *dp = dump ? dump + p * (size_t)r.max_bounces * 16 : nullptr;
                    V3 c = trace_pixel(s, u, u.frameIndex + r.sample_offset, r.seeds[p], x, y, r.max_bounces, brute, cnts[tid], dp, r.materials);
                    float *a = &r.accum[p * 4];
                    if (u.frameIndex > 0) {                                        // :395-401
                        float fi = (float)u.frameIndex;
                        V3 prev = v3(a[0], a[1], a[2]) * fi;
                        c = c + prev;
                        float den = (float)(u.frameIndex + 1);
                        c = v3(c.x / den, c.y / den, c.z / den);
                    }
                    a[0] = c.x; a[1] = c.y; a[2] = c.z; a[3] = 1.0f;              // :403
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
        work(0);
        for (auto &t : th) t.join();
        for (auto &c : cnts) { r.total.closest += c.closest; r.total.shadow += c.shadow; }
        r.frameIndex++;
    }
}

}  // namespace

// ================================================================== C interface (ctypes)
extern "C" {

void *orc_scene_create() { return new Scene(); }
void orc_scene_destroy(void *s) { delete (Scene *)s; }
int orc_scene_add_mesh(void *sp, const float *pos, const float *nrm, size_t nverts, const float *xf16) {
    Scene *s = (Scene *)sp; Mesh m;
    m.pos.resize(nverts); m.nrm.resize(nverts);
    for (size_t i = 0; i < nverts; i++) { m.pos[i] = v3(pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]); m.nrm[i] = v3(nrm[i * 3], nrm[i * 3 + 1], nrm[i * 3 + 2]); }
    memcpy(&m.xf, xf16, 64);
    m.xf.c[0][3] = m.xf.c[1][3] = m.xf.c[2][3] = 0; m.xf.c[3][3] = 1;     // drop_last_row (Utilities.swift:92-101)
    s->meshes.push_back(std::move(m)); s->committed = false;
    return (int)s->meshes.size() - 1;
}
int orc_mesh_add_submesh(void *sp, int mesh, const uint32_t *idx, size_t ntris, const MRTMaterial *mat) {
    Scene *s = (Scene *)sp; if (mesh < 0 || mesh >= (int)s->meshes.size()) return -1;
    Mesh &m = s->meshes[mesh];
    m.sub_idx.emplace_back(idx, idx + ntris * 3); m.sub_mat.push_back(*mat); s->committed = false;
    return (int)m.sub_idx.size() - 1;
}
int orc_scene_add_instance(void *sp, int source, const float *xf16) {
    Scene *s = (Scene *)sp; if (source < 0 || source >= (int)s->meshes.size()) return -1;
    Mesh m; m.source = s->meshes[source].source >= 0 ? s->meshes[source].source : source;
    memcpy(&m.xf, xf16, 64);
    m.xf.c[0][3] = m.xf.c[1][3] = m.xf.c[2][3] = 0; m.xf.c[3][3] = 1;
    s->meshes.push_back(std::move(m)); s->committed = false;
    return (int)s->meshes.size() - 1;
}
void orc_scene_set_instancing(void *sp, int on) { Scene *s = (Scene *)sp; s->instancing = on != 0; s->committed = false; }
void orc_scene_set_transform(void *sp, int mesh, const float *xf16) {
    Scene *s = (Scene *)sp; if (mesh < 0 || mesh >= (int)s->meshes.size()) return;
    memcpy(&s->meshes[mesh].xf, xf16, 64);
    M4 &x = s->meshes[mesh].xf; x.c[0][3] = x.c[1][3] = x.c[2][3] = 0; x.c[3][3] = 1; s->committed = false;
}
void orc_scene_set_lights(void *sp, const MRTLight *l, int n) { Scene *s = (Scene *)sp; s->lights.assign(l, l + n); }
void orc_scene_commit(void *sp) { scene_commit(*(Scene *)sp); }
uint64_t orc_scene_triangles(void *sp) { return ((Scene *)sp)->total_tris; }
uint64_t orc_scene_nodes(void *sp) { return ((Scene *)sp)->nodes.size(); }

static void fill_isect(const Scene *s, const Hit &h, MRTIntersection *o) {
    memset(o, 0, sizeof *o);
    if (h.gid == 0xFFFFFFFFu) { o->type = 0; o->distance = -1.0f; o->instance_id = o->geometry_id = o->primitive_id = -1; return; }
    const TriRef &t = s->refs[h.gid];
    o->type = 1; o->distance = h.t; o->instance_id = t.inst; o->geometry_id = t.geom; o->primitive_id = t.prim;
    o->u = h.U / h.ad; o->v = h.V / h.ad;
}
void orc_intersect_closest(void *sp, const MRTRay *rays, size_t n, MRTIntersection *out, int brute) {
    Scene *s = (Scene *)sp;
    for (size_t i = 0; i < n; i++) {
        V3 o = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]), d = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
        Hit h = scene_closest(*s, o, d, rays[i].min_distance, rays[i].max_distance, brute != 0);
        fill_isect(s, h, &out[i]);
    }
}
void orc_intersect_any(void *sp, const MRTRay *rays, size_t n, int32_t *occ, int brute) {
    Scene *s = (Scene *)sp;
    for (size_t i = 0; i < n; i++) {
        V3 o = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]), d = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
        occ[i] = scene_any(*s, o, d, rays[i].min_distance, rays[i].max_distance, brute != 0) ? 1 : 0;
    }
}

void *orc_renderer_create(void *scene, int w, int h, uint32_t seed, int max_bounces) {
    Renderer *r = new Renderer();
    r->scene = (Scene *)scene; r->w = w; r->h = h; r->seed = seed; r->max_bounces = max_bounces;
    default_camera(w, h, &r->cam);
    r->seeds.resize((size_t)w * h);
    for (size_t i = 0; i < r->seeds.size(); i++) r->seeds[i] = seed_hash(seed, (uint32_t)i);
    r->accum.assign((size_t)w * h * 4, 0.0f);
    return r;
}
void orc_renderer_destroy(void *r) { delete (Renderer *)r; }
void orc_renderer_set_camera(void *rp, const MRTCamera *c) { ((Renderer *)rp)->cam = *c; }
void orc_renderer_set_shard(void *rp, int rank, int world) { Renderer *r = (Renderer *)rp; r->shard_rank = rank; r->shard_world = world; }
void orc_renderer_set_frame_index(void *rp, uint32_t fi) { ((Renderer *)rp)->frameIndex = fi; }
void orc_renderer_set_sample_offset(void *rp, uint32_t so) { ((Renderer *)rp)->sample_offset = so; }
void orc_renderer_set_materials(void *rp, int on) { ((Renderer *)rp)->materials = on != 0; }
void orc_renderer_set_accum(void *rp, const float *rgba) { Renderer *r = (Renderer *)rp; memcpy(r->accum.data(), rgba, r->accum.size() * 4); }
// dump: NULL or w*h*max_bounces*16 floats (per-bounce stage records; only the last frame's survive)
void orc_renderer_render(void *rp, int nframes, int nthreads, int brute, float *dump) {
    Renderer *r = (Renderer *)rp;
    if (nthreads < 1) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    render_frames(*r, nframes, nthreads, brute != 0, dump);
}
void orc_renderer_read_accum(void *rp, float *rgba) { Renderer *r = (Renderer *)rp; memcpy(rgba, r->accum.data(), r->accum.size() * 4); }
void orc_renderer_counters(void *rp, uint64_t *closest, uint64_t *shadow) { Renderer *r = (Renderer *)rp; *closest = r->total.closest; *shadow = r->total.shadow; }
// Shaders.metal:39-52 + the blit's vertical flip (:35): top row first, RGBA8
void orc_tonemap_rgba8(const float *rgba, int w, int h, uint8_t *out) {
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        const float *a = &rgba[((size_t)(h - 1 - y) * w + x) * 4]; uint8_t *o = &out[((size_t)y * w + x) * 4];
        for (int k = 0; k < 3; k++) { float c = a[k]; c = c / (1.0f + c); c = saturate(c); o[k] = (uint8_t)(c * 255.0f + 0.5f); }
        o[3] = 255;
    }
}

// ---- scalar helpers for known-answer tests
float orc_halton(int i, int d) { return halton(i, d); }
void orc_sincos_2pi(float u, float *s, float *c) { sincos_2pi(u, s, c); }
void orc_hemisphere(float ux, float uy, float *out3) { V3 v = sample_cosine_hemisphere(ux, uy); out3[0] = v.x; out3[1] = v.y; out3[2] = v.z; }
void orc_align(const float *s3, const float *n3, float *out3) { V3 v = align_hemisphere(v3(s3[0], s3[1], s3[2]), v3(n3[0], n3[1], n3[2])); out3[0] = v.x; out3[1] = v.y; out3[2] = v.z; }
uint32_t orc_seed_hash(uint32_t seed, uint32_t idx) { return seed_hash(seed, idx); }
void orc_make_transform(const float *p, const float *r, float s, float *out16) { M4 m = m4_trs(v3(p[0], p[1], p[2]), v3(r[0], r[1], r[2]), s); memcpy(out16, &m, 64); }
void orc_default_camera(int w, int h, MRTCamera *c) { default_camera(w, h, c); }
// sampleAreaLight (Raytracing.metal:94-128) on raw inputs
void orc_sample_area_light(const MRTLight *L, const float *u2, const float *pos3, float *dir3, float *col3, float *dist) {
    float ax = u2[0] * 2.0f - 1.0f, ay = u2[1] * 2.0f - 1.0f;
    V3 P = v3(pos3[0], pos3[1], pos3[2]);
    V3 sp = (v3(L->position) + v3(L->right) * ax) + v3(L->up) * ay;
    V3 ldir = sp - P; float ldist = length3(ldir);
    float inv = 1.0f / (ldist > 1e-3f ? ldist : 1e-3f);
    ldir = ldir * inv;
    V3 lcol = v3(L->color) * (inv * inv);
    lcol = lcol * saturate(dot3(neg(ldir), v3(L->forward)));
    dir3[0] = ldir.x; dir3[1] = ldir.y; dir3[2] = ldir.z; col3[0] = lcol.x; col3[1] = lcol.y; col3[2] = lcol.z; *dist = ldist;
}

}  // extern "C"
