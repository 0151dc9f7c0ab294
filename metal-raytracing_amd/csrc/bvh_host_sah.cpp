// bvh_host_sah.cpp — builder = 2: a top-down binned-SAH binary tree built on the HOST (all cores), as the quality yardstick for the device
// builders (0 = Karras LBVH, 1 = PLOC).  It produces only the TOPOLOGY over the build references (triangles, or pieces of pre-split slivers): the
// leaf order and left / right / parent per node, in the numbering the device pipeline expects (internal nodes 0 .. n-2, leaves n-1+j for position j of
// the leaf order); boxes, SAH collapse, rope links and the 8-wide layout come from the same device kernels as for the other builders
// (bvh_build.hip: k_refit ... k_wide_level), so that the A/B isolates what a better topology is worth.  Replaces (as the others do) the opaque
// MTLAccelerationStructure build of Renderer.swift:184-214.  Split search: 32 bins over the centroid bounds on each axis, SAH = area x count of both
// sides; ranges of <= 8 references are split by an exact sweep along the best axis; degenerate centroid bounds fall back to a median split.
#include "scene_device.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace mrt {
namespace {

struct B3 { float lo[3], hi[3]; };
inline void b_init(B3 &b) { for (int k = 0; k < 3; k++) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; } }
inline void b_add(B3 &b, const float *lo, const float *hi) { for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], lo[k]); b.hi[k] = std::max(b.hi[k], hi[k]); } }
inline void b_add(B3 &b, const B3 &o) { b_add(b, o.lo, o.hi); }
// bin of a centroid: clamped as a float first — a NaN or an out-of-range product (non-finite boxes, an extent that overflows) must not reach the float -> int conversion
inline int bin_of(float c, float lo, float scale, int bins) { const float f = (c - lo) * scale; return f >= (float)(bins - 1) ? bins - 1 : f > 0.0f ? (int)f : 0; }
inline float b_area(const B3 &b) { const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2]; return x < 0.0f ? 0.0f : x * y + y * z + z * x; }

struct Builder {
    const float4 *lo, *hi;                  // per reference
    uint32_t n;
    std::vector<uint32_t> &order, &left, &right, &parent;
    std::vector<float> cx, cy, cz;          // centroids (2 x centre)
    static constexpr int BINS = 32;

    const float *cen(int a) const { return a == 0 ? cx.data() : a == 1 ? cy.data() : cz.data(); }

    // builds the subtree over order[b, e) and returns its node id
    uint32_t build(uint32_t b, uint32_t e, uint32_t par, int depth, std::vector<std::thread> *pool) {
        const uint32_t cnt = e - b;
        if (cnt == 1) { const uint32_t id = n - 1 + b; parent[id] = par; return id; }
        float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t i = b; i < e; i++) {
            const uint32_t r = order[i];
            clo[0] = std::min(clo[0], cx[r]); chi[0] = std::max(chi[0], cx[r]); clo[1] = std::min(clo[1], cy[r]); chi[1] = std::max(chi[1], cy[r]); clo[2] = std::min(clo[2], cz[r]); chi[2] = std::max(chi[2], cz[r]);
        }
        uint32_t mid = b + cnt / 2; int axis = 0; bool found = false;
        if (cnt <= 8) {                  // exact sweep
            float best = 3.0e38f;
            for (int a = 0; a < 3; a++) {
                if (!(chi[a] > clo[a])) continue;
                const float *c = cen(a);
                uint32_t tmp[8]; for (uint32_t i = 0; i < cnt; i++) tmp[i] = order[b + i];
                std::sort(tmp, tmp + cnt, [&](uint32_t x, uint32_t y) { return c[x] < c[y] || (c[x] == c[y] && x < y); });
                float ra[8]; B3 acc; b_init(acc);
                for (uint32_t i = cnt; i-- > 1;) { b_add(acc, &lo[tmp[i]].x, &hi[tmp[i]].x); ra[i] = b_area(acc); }
                b_init(acc);
                for (uint32_t i = 1; i < cnt; i++) {
                    b_add(acc, &lo[tmp[i - 1]].x, &hi[tmp[i - 1]].x);
                    const float cost = b_area(acc) * (float)i + ra[i] * (float)(cnt - i);
                    if (cost < best) { best = cost; axis = a; mid = b + i; found = true; }
                }
            }
            if (found) { const float *c = cen(axis); std::sort(order.begin() + b, order.begin() + e, [&](uint32_t x, uint32_t y) { return c[x] < c[y] || (c[x] == c[y] && x < y); }); }
        } else {
            float best = 3.0e38f; int bbin = -1;
            for (int a = 0; a < 3; a++) {
                const float ext = chi[a] - clo[a];
                if (!(ext > 0.0f)) continue;
                const float *c = cen(a);
                const float scale = (float)BINS / ext;
                B3 bb[BINS]; uint32_t bc[BINS];
                for (int k = 0; k < BINS; k++) { b_init(bb[k]); bc[k] = 0; }
                for (uint32_t i = b; i < e; i++) {
                    const uint32_t r = order[i];
                    const int k = bin_of(c[r], clo[a], scale, BINS);
                    b_add(bb[k], &lo[r].x, &hi[r].x); bc[k]++;
                }
                float ra[BINS]; uint32_t rc[BINS]; B3 acc; b_init(acc); uint32_t c2 = 0;
                for (int k = BINS - 1; k >= 1; k--) { b_add(acc, bb[k]); c2 += bc[k]; ra[k] = b_area(acc); rc[k] = c2; }
                b_init(acc); c2 = 0;
                for (int k = 1; k < BINS; k++) {
                    b_add(acc, bb[k - 1]); c2 += bc[k - 1];
                    if (c2 == 0 || rc[k] == 0) continue;
                    const float cost = b_area(acc) * (float)c2 + ra[k] * (float)rc[k];
                    if (cost < best) { best = cost; axis = a; bbin = k; }
                }
            }
            if (bbin >= 0) {
                const float *c = cen(axis);
                const float scale = (float)BINS / (chi[axis] - clo[axis]);
                auto it = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t r) { return bin_of(c[r], clo[axis], scale, BINS) < bbin; });
                mid = (uint32_t)(it - order.begin());
                found = mid > b && mid < e;
            }
        }
        if (!found) {                    // all centroids equal (or a failed partition): median split in the current order
            mid = b + cnt / 2;
        }
        // the internal node that splits [b, e) between positions mid - 1 and mid is node mid - 1: every split position occurs once in the tree, so the ids are
        // 0 .. n - 2 whatever the threads' timing (an atomic counter used to hand them out in arrival order: a different numbering from run to run)
        const uint32_t id = mid - 1;
        parent[id] = par;
        uint32_t l, r;
        if (pool && depth < 6 && cnt > 4096) {       // the top of the tree fans out over threads: both halves keep fanning out, 2^6 threads at most
            uint32_t lr = 0; bool spawned = false;
            std::thread th;
            try { th = std::thread([&, this] { lr = build(b, mid, id, depth + 1, pool); }); spawned = true; } catch (...) {}      // no thread to be had: this half is built here
            struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{th};      // joined on every path out of this scope, also when the other half throws
            r = build(mid, e, id, depth + 1, pool);
            if (spawned) { th.join(); l = lr; } else l = build(b, mid, id, depth + 1, pool);
        } else { l = build(b, mid, id, depth + 1, pool); r = build(mid, e, id, depth + 1, pool); }
        left[id] = l; right[id] = r;
        return id;
    }
};

}  // namespace

// lo / hi: n reference boxes (host copies).  Out: order[n] (leaf position -> reference), left / right [n - 1] and parent [2n - 1] in the pipeline's numbering.
void host_sah_topology(const float4 *lo, const float4 *hi, uint32_t n, std::vector<uint32_t> &order, std::vector<uint32_t> &left, std::vector<uint32_t> &right, std::vector<uint32_t> &parent) {
    order.resize(n); left.assign(n > 1 ? n - 1 : 1, 0); right.assign(n > 1 ? n - 1 : 1, 0); parent.assign(2 * (size_t)n - 1, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    Builder bl{lo, hi, n, order, left, right, parent, {}, {}, {}};
    bl.cx.resize(n); bl.cy.resize(n); bl.cz.resize(n);
    for (uint32_t i = 0; i < n; i++) { bl.cx[i] = lo[i].x + hi[i].x; bl.cy[i] = lo[i].y + hi[i].y; bl.cz[i] = lo[i].z + hi[i].z; }
    std::vector<std::thread> pool;
    if (n >= 1) bl.build(0, n, 0xFFFFFFFFu, 0, &pool);
}

}  // namespace mrt
