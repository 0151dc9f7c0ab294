// two_level.hip — instanced scenes: one BLAS per distinct mesh + a TLAS over the instances (scene option instancing = 1).
//
// Replaces the reference's instance acceleration structure: MTLAccelerationStructureInstanceDescriptor per mesh (transform,
// accelerationStructureIndex; Renderer.swift:193-203) over an array of primitive acceleration structures (:184-192, :205-213),
// and the compaction pass of Utilities.swift:65-84 (here: each BLAS is built into a scratch allocation and only its surviving
// nodes are copied into the shared arrays).  The reference never points two instances at one primitive structure
// (DragonScene.swift:19-20 loads the sphere twice), but its API is instance-based; mrt_scene_add_instance makes the sharing
// explicit, BASELINE.json configs[4] ("dragon.obj at 4x instancing") uses it.
//
//   BLAS   the flattened-scene pipeline of bvh_build.hip (Morton sort -> PLOC -> SAH collapse -> rope nodes + packets) run on ONE mesh
//          under the identity transform, i.e. in object space; triangle ids are local to the mesh.
//   TLAS   a rope tree over the instances' world boxes (the 8 corners of the BLAS root box through the instance matrix), built on the
//          host: median split of the box centres along the widest axis, leaves of up to two instances.  Instance counts are small
//          (<= 65 535 by the ABI) and the build is microseconds; a transform change rebuilds ONLY this tree and the 80-byte
//          instance rows (update_tlas) — the BLASes are never touched ("refit" of an animated scene).
//   8-wide every BLAS also gets the compressed 8-wide layout (bvh_build.hip k_wide_level), all of them in ONE node array with absolute child /
//          packet indices, behind `tlas_wcap` slots reserved for an 8-wide TLAS whose leaf children are single instances (WideTlasBuilder,
//          same node format, host-built, rewritten in place by update_tlas).  The stream traversal of traverse_wide.h walks both levels
//          with one loop and one LDS stack; the rope layout stays for the query kernels and as the A/B path (wide_bounce = 0).
#include "scene_device.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>

namespace mrt {
namespace {

// inverse of the affine map p -> A p + t given as a column-major 4x4 (last row 0 0 0 1): rows of [A^-1 | -A^-1 t], evaluated in double with
// this exact operation order (the CPU oracle restates it: the instance rows must agree bit for bit) and rounded to float once.
bool invert_affine(const float *xf, float rows[3][4]) {
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

struct Box { float lo[3], hi[3]; };

// world box of an instance: the corners of its BLAS root box through the object->world matrix; padded like the leaf boxes of
// bvh_build.hip (1e-5 |coord| + 1e-6) plus the rounding of the ray's trip into object space
Box instance_box(const float *xf, const float *lo, const float *hi) {
    Box b; for (int k = 0; k < 3; k++) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; }
    for (int c = 0; c < 8; c++) {
        const double p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
        for (int k = 0; k < 3; k++) {
            const double w = (double)xf[k] * p[0] + (double)xf[4 + k] * p[1] + (double)xf[8 + k] * p[2] + (double)xf[12 + k];
            b.lo[k] = std::min(b.lo[k], (float)w); b.hi[k] = std::max(b.hi[k], (float)w);
        }
    }
    for (int k = 0; k < 3; k++) {
        const float m = std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])), e = 4e-5f * m + 4e-6f;
        b.lo[k] -= e; b.hi[k] += e;
    }
    return b;
}

// Host TLAS: preorder rope nodes in the 64-byte layout of scene_device.h.  The left child is entered first for every octant (near
// mask 0), so all eight escape links of a node are equal; leaves are ranges of `order`.
struct TlasBuilder {
    const std::vector<Box> &boxes;
    std::vector<uint32_t> order;          // instance ids, leaf order
    std::vector<float4> nodes;            // 4 per node
    int depth = 0;
    explicit TlasBuilder(const std::vector<Box> &b) : boxes(b) {}
    static float4 f4(float x, float y, float z, uint32_t w) { float f; memcpy(&f, &w, 4); return make_float4(x, y, z, f); }
    // builds the subtree over order[first, first + count) and returns its node index; `esc` = node to continue with afterwards
    uint32_t build(uint32_t first, uint32_t count, uint32_t esc, int d) {
        depth = std::max(depth, d);
        Box b; for (int k = 0; k < 3; k++) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; }
        float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t i = first; i < first + count; i++) {
            const Box &q = boxes[order[i]];
            for (int k = 0; k < 3; k++) {
                b.lo[k] = std::min(b.lo[k], q.lo[k]); b.hi[k] = std::max(b.hi[k], q.hi[k]);
                const float c = 0.5f * (q.lo[k] + q.hi[k]); clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c);
            }
        }
        const uint32_t me = (uint32_t)(nodes.size() / 4);
        nodes.resize(nodes.size() + 4);
        uint32_t a, bb;
        if (count <= 2) { a = NODE_LEAF | first; bb = count; }
        else {
            int ax = 0; if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1; if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
            const uint32_t half = count / 2;
            std::nth_element(order.begin() + first, order.begin() + first + half, order.begin() + first + count, [&](uint32_t x, uint32_t y) {
                const float cx = boxes[x].lo[ax] + boxes[x].hi[ax], cy = boxes[y].lo[ax] + boxes[y].hi[ax];
                return cx < cy || (cx == cy && x < y);
            });
            // preorder: the left subtree starts at me + 1; the right subtree's index is known once the left one is built
            const uint32_t left = me + 1;
            // the left subtree escapes to the right sibling, whose index = me + 1 + size(left subtree): build left with a placeholder, patch after
            const size_t before = nodes.size();
            build(first, half, 0xFFFFFFFEu, d + 1);
            const uint32_t right = (uint32_t)(nodes.size() / 4);
            for (size_t n = before; n < nodes.size(); n += 4)       // patch the placeholder links of the left subtree
                for (int q = 2; q < 4; q++) {
                    uint32_t w[4]; memcpy(w, &nodes[n + q], 16);
                    for (int k = 0; k < 4; k++) if (w[k] == 0xFFFFFFFEu) w[k] = right;
                    memcpy(&nodes[n + q], w, 16);
                }
            build(first + half, count - half, esc, d + 1);
            a = left; bb = right;                                      // near mask 0: left first
        }
        nodes[4 * (size_t)me + 0] = f4(b.lo[0], b.lo[1], b.lo[2], a);
        nodes[4 * (size_t)me + 1] = f4(b.hi[0], b.hi[1], b.hi[2], bb);
        float e; memcpy(&e, &esc, 4);
        nodes[4 * (size_t)me + 2] = make_float4(e, e, e, e);
        nodes[4 * (size_t)me + 3] = make_float4(e, e, e, e);
        return me;
    }
};


// 8-wide TLAS in the node format of scene_device.h: a median-split binary tree over the instance boxes (leaves = ONE instance), collapsed
// greedily (largest surface area first) into nodes of up to eight children, BFS-numbered so that a node's internal children are contiguous;
// child slots by octant of the child centre, boxes quantised outwards to 8 bits on the node's power-of-two grid — the host restatement of
// k_wide_level.  A leaf child's "packet" tri_base + offset is an entry of `order` (instance ids).
struct WideTlasBuilder {
    const std::vector<Box> &boxes;
    struct BNode { Box b; int left = -1, right = -1, inst = -1; };
    std::vector<BNode> bn;
    std::vector<uint32_t> order;          // instance ids, leaf-child order
    std::vector<float4> nodes;            // 5 per wide node
    int depth = 0;
    explicit WideTlasBuilder(const std::vector<Box> &b) : boxes(b) {}
    static float area(const Box &q) { const float x = q.hi[0] - q.lo[0], y = q.hi[1] - q.lo[1], z = q.hi[2] - q.lo[2]; return x * y + y * z + z * x; }
    int binary(std::vector<uint32_t> &ids, uint32_t first, uint32_t count) {
        BNode n; for (int k = 0; k < 3; k++) { n.b.lo[k] = 3.0e38f; n.b.hi[k] = -3.0e38f; }
        float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t i = first; i < first + count; i++) {
            const Box &q = boxes[ids[i]];
            for (int k = 0; k < 3; k++) {
                n.b.lo[k] = std::min(n.b.lo[k], q.lo[k]); n.b.hi[k] = std::max(n.b.hi[k], q.hi[k]);
                const float c = 0.5f * (q.lo[k] + q.hi[k]); clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c);
            }
        }
        const int me = (int)bn.size(); bn.push_back(n);
        if (count == 1) { bn[me].inst = (int)ids[first]; return me; }
        int ax = 0; if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1; if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
        const uint32_t half = count / 2;
        std::nth_element(ids.begin() + first, ids.begin() + first + half, ids.begin() + first + count, [&](uint32_t x, uint32_t y) {
            const float cx = boxes[x].lo[ax] + boxes[x].hi[ax], cy = boxes[y].lo[ax] + boxes[y].hi[ax];
            return cx < cy || (cx == cy && x < y);
        });
        const int l = binary(ids, first, half), r = binary(ids, first + half, count - half);
        bn[me].left = l; bn[me].right = r;
        return me;
    }
    static float4 f4u(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { float f[4]; uint32_t u[4] = {a, b, c, d}; memcpy(f, u, 16); return make_float4(f[0], f[1], f[2], f[3]); }
    void build(std::vector<uint32_t> ids) {
        if (ids.empty()) return;
        const int root = binary(ids, 0, (uint32_t)ids.size());
        std::vector<int> level{root};                 // binary nodes that become the wide nodes of this level
        uint32_t base = 0;
        while (!level.empty()) {
            depth++;
            std::vector<int> next;
            const uint32_t next_base = base + (uint32_t)level.size();
            nodes.resize(nodes.size() + 5 * level.size());
            for (size_t i = 0; i < level.size(); i++) {
                const BNode &f = bn[(size_t)level[i]];
                int ch[8], nch = 0;
                if (f.inst >= 0) ch[nch++] = level[i];
                else {
                    ch[nch++] = f.left; ch[nch++] = f.right;
                    while (nch < 8) {
                        int best = -1; float ba = -1.0f;
                        for (int k = 0; k < nch; k++) if (bn[(size_t)ch[k]].inst < 0) { const float a = area(bn[(size_t)ch[k]].b); if (a > ba) { ba = a; best = k; } }
                        if (best < 0) break;
                        const int c = ch[best];
                        ch[best] = bn[(size_t)c].left; ch[nch++] = bn[(size_t)c].right;
                    }
                }
                int child_in_slot[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
                const float cx = f.b.lo[0] + f.b.hi[0], cy = f.b.lo[1] + f.b.hi[1], cz = f.b.lo[2] + f.b.hi[2];
                for (int k = 0; k < nch; k++) {
                    const Box &q = bn[(size_t)ch[k]].b;
                    const int pref = ((q.lo[0] + q.hi[0]) > cx ? 1 : 0) | ((q.lo[1] + q.hi[1]) > cy ? 2 : 0) | ((q.lo[2] + q.hi[2]) > cz ? 4 : 0);
                    int bs = -1, bd = 99;
                    for (int sl = 0; sl < 8; sl++) if (child_in_slot[sl] < 0) { const int dd = __builtin_popcount((unsigned)(sl ^ pref)); if (dd < bd) { bd = dd; bs = sl; } }
                    child_in_slot[bs] = k;
                }
                uint32_t eb[3]; float step[3], inv_step[3];
                for (int a = 0; a < 3; a++) {
                    const float sdiv = (f.b.hi[a] - f.b.lo[a]) / 255.0f;
                    uint32_t bits; memcpy(&bits, &sdiv, 4);
                    uint32_t e = (bits >> 23) + ((bits & 0x7FFFFFu) ? 1u : 0u);
                    e = std::min(254u, std::max(1u, e));
                    eb[a] = e;
                    const uint32_t sb = e << 23, ib = (254u - e) << 23; memcpy(&step[a], &sb, 4); memcpy(&inv_step[a], &ib, 4);
                }
                uint32_t q[6][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}}, meta[2] = {0, 0}, imask = 0, off_t = 0;
                const uint32_t my_i = next_base + (uint32_t)next.size(), my_t = (uint32_t)order.size();
                for (int sl = 0; sl < 8; sl++) {
                    const int k = child_in_slot[sl];
                    uint32_t ql[3] = {255, 255, 255}, qh[3] = {0, 0, 0};
                    if (k >= 0) {
                        const BNode &c = bn[(size_t)ch[k]];
                        for (int a = 0; a < 3; a++) {
                            float fl = std::floor((c.b.lo[a] - f.b.lo[a]) * inv_step[a]), fh = std::ceil((c.b.hi[a] - f.b.lo[a]) * inv_step[a]);
                            fl = std::min(std::max(fl, 0.0f), 255.0f); fh = std::min(std::max(fh, 0.0f), 255.0f);
                            if (f.b.lo[a] + fl * step[a] > c.b.lo[a] && fl > 0.0f) fl -= 1.0f;
                            if (f.b.lo[a] + fh * step[a] < c.b.hi[a] && fh < 255.0f) fh += 1.0f;
                            ql[a] = (uint32_t)fl; qh[a] = (uint32_t)fh;
                        }
                        if (c.inst >= 0) { meta[sl >> 2] |= ((1u << 5) | off_t) << (8 * (sl & 3)); order.push_back((uint32_t)c.inst); off_t++; }
                        else { imask |= 1u << sl; next.push_back(ch[k]); }
                    }
                    for (int a = 0; a < 3; a++) { q[a][sl >> 2] |= ql[a] << (8 * (sl & 3)); q[3 + a][sl >> 2] |= qh[a] << (8 * (sl & 3)); }
                }
                const size_t w = 5 * (size_t)(base + i);
                float ew; const uint32_t ewb = ((eb[0] - 127u) & 0xFFu) | (((eb[1] - 127u) & 0xFFu) << 8) | (((eb[2] - 127u) & 0xFFu) << 16) | (imask << 24); memcpy(&ew, &ewb, 4);
                nodes[w + 0] = make_float4(f.b.lo[0], f.b.lo[1], f.b.lo[2], ew);
                nodes[w + 1] = f4u(my_i, my_t, meta[0], meta[1]);
                nodes[w + 2] = f4u(q[0][0], q[0][1], q[1][0], q[1][1]);
                nodes[w + 3] = f4u(q[2][0], q[2][1], q[3][0], q[3][1]);
                nodes[w + 4] = f4u(q[4][0], q[4][1], q[5][0], q[5][1]);
            }
            base = next_base;
            level.swap(next);
        }
    }
};

// a BLAS's 8-wide nodes after the copy into the shared array: child / packet bases become absolute
__global__ void k_relocate_wide(float4 *__restrict__ wnodes, uint32_t n, uint32_t node_off, uint32_t packet_off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 n1 = wnodes[WNODE_STRIDE * (size_t)i + 1];
    n1.x = __uint_as_float(__float_as_uint(n1.x) + node_off); n1.y = __uint_as_float(__float_as_uint(n1.y) + packet_off);
    wnodes[WNODE_STRIDE * (size_t)i + 1] = n1;
}

// tri_packet[ts_base + t] = the packet (absolute index in wpackets) of triangle t of a BLAS: the inverse of the packets' id words
__global__ void k_packet_map(const float4 *__restrict__ wpackets, uint32_t packet_base, uint32_t nt, uint32_t ts_base, uint32_t *__restrict__ tri_packet) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nt) return;
    const uint32_t local = __float_as_uint(wpackets[WPK * (size_t)(packet_base + p)].w);
    if (local < nt) tri_packet[ts_base + local] = packet_base + p;
}

// The per-node checks of validate_layout for the nodes [first, last) of the 8-wide array that are NOT TLAS slots, on the device: the first violation (lowest node) is left in
// *err as node << 8 | code.  (The host loop over 73 K downloaded nodes was 2.2 of DragonScene's 9.1 ms commit; the TLAS slots, a handful of nodes whose leaves name instances, stay on the host.)
enum { V_OK = 0, V_ORDER = 1, V_RANGE = 2, V_INTO_TLAS = 3, V_BOTH = 4, V_MASK = 5, V_PACKETS = 6 };
__global__ void k_validate_wide(const float4 *__restrict__ wnodes, uint32_t first, uint32_t last, uint32_t num_wnodes, uint32_t tlas_wcap, uint32_t two_level, unsigned long long packets, unsigned long long *__restrict__ err) {
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last) return;
    const float4 n0 = wnodes[WNODE_STRIDE * (size_t)i], n1 = wnodes[WNODE_STRIDE * (size_t)i + 1];
    const uint32_t imask = __float_as_uint(n0.w) >> 24, child_base = __float_as_uint(n1.x), tri_base = __float_as_uint(n1.y), meta[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)};
    uint32_t code = V_OK;
    const uint32_t ninner = (uint32_t)__popc(imask);
    if (ninner) {
        if (child_base <= i) code = V_ORDER;
        else if ((unsigned long long)child_base + ninner > num_wnodes) code = V_RANGE;
        else if (two_level && child_base < tlas_wcap) code = V_INTO_TLAS;
    }
    for (int sl = 0; sl < 8 && code == V_OK; sl++) {
        const uint32_t m = (meta[sl >> 2] >> (8 * (sl & 3))) & 0xFFu, cnt = m >> 5, off = m & 31u;
        if (!cnt) continue;
        if ((imask >> sl) & 1u) code = V_BOTH;
        else if (off + cnt > 32u) code = V_MASK;
        else if ((unsigned long long)tri_base + off + cnt > packets) code = V_PACKETS;
    }
    if (code != V_OK) atomicMin(err, ((unsigned long long)i << 8) | code);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// Commit-time validation of the 8-wide layout (and, for two-level scenes, of the instance rows, the index arrays and the rope TLAS): every
// index a traversal kernel will follow must lie inside its array, and every internal child must come after its parent (BFS numbering: no
// cycles, so every walk ends).  A work-in-progress build of the two-level path once ended in "Memory access fault by GPU" at the first
// query after a commit: an index outside its array is now refused HERE, with a message, instead of being dereferenced by a
// kernel.  Cost: one download of the nodes checked (80 B each) and a host loop — about a millisecond per 100 K nodes, once per commit; a
// TLAS-only update re-checks only the TLAS slots, the instance rows and the index arrays (microseconds).
int validate_layout(const DeviceScene &sc, hipStream_t stream, bool tlas_only, const float4 *wnodes_override) {
    const float4 *const wnodes = wnodes_override ? wnodes_override : sc.wnodes.p;
    auto bad = [&](const std::string &what) { set_error("scene layout validation failed: " + what); return MRT_ERR_STATE; };
    const bool two = sc.num_inst > 0;
    const uint32_t NW = sc.num_wnodes;
    const size_t packets = sc.wpackets.p ? sc.wpackets.n / WPK : 0;
    std::vector<uint32_t> wtl, tl;
    if (two) {
        const size_t I = sc.num_inst;
        if (sc.h_inst.size() != I) return bad("instance table size");
        const size_t rope_nodes = sc.bpackets_offset / 4, rope_packets = sc.bnodes.n >= sc.bpackets_offset ? (sc.bnodes.n - sc.bpackets_offset) / 3 : 0;
        for (size_t i = 0; i < I; i++) {
            const InstanceDev &d = sc.h_inst[i];
            const std::string who = "instance " + std::to_string(i);
            if (d.ntri == 0) continue;
            if ((size_t)d.node_base >= std::max<size_t>(rope_nodes, 1)) return bad(who + ": node_base outside bnodes");
            if ((size_t)d.packet_base + d.ntri > rope_packets) return bad(who + ": packet range outside bpackets");
            if ((size_t)d.ts_base + d.ntri > sc.tri_shade.n) return bad(who + ": shading records outside tri_shade");
            if ((size_t)d.vbase >= sc.normals.n) return bad(who + ": vbase outside normals");
            if (NW) {
                if (d.ntri <= 8u) { if ((size_t)d.packet_base + d.ntri > packets) return bad(who + ": inlined packets outside wpackets"); }
                else if (d.wroot < sc.tlas_wcap || d.wroot >= NW) return bad(who + ": wroot outside the BLAS part of wnodes");
            }
            for (int r = 0; r < 3; r++) { const float4 q = d.w2o[r]; if (!(std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w))) return bad(who + ": world->object row is not finite"); }
        }
        tl.resize(sc.tlas_index.n);
        MRT_HIP(hipMemcpyAsync(tl.data(), sc.tlas_index.p, tl.size() * 4, hipMemcpyDeviceToHost, stream));
        if (NW) { wtl.resize(sc.wtlas_index.n); MRT_HIP(hipMemcpyAsync(wtl.data(), sc.wtlas_index.p, wtl.size() * 4, hipMemcpyDeviceToHost, stream)); }
        // rope TLAS (64-byte nodes: lo | a, hi | b, esc[8])
        const size_t tn = sc.stats.bvh_nodes;
        std::vector<float4> rn(4 * tn);
        if (tn) MRT_HIP(hipMemcpyAsync(rn.data(), sc.nodes.p, rn.size() * 16, hipMemcpyDeviceToHost, stream));
        MRT_HIP(hipStreamSynchronize(stream));
        // (index arrays are allocated with at least one element; entries past the live instances are never referenced by a node)
        for (size_t k = 0; k < tn; k++) {
            uint32_t a, b, esc[8]; memcpy(&a, &rn[4 * k].w, 4); memcpy(&b, &rn[4 * k + 1].w, 4); memcpy(esc, &rn[4 * k + 2], 32);
            const std::string who = "TLAS rope node " + std::to_string(k);
            if (a & NODE_LEAF) { const size_t first = a & 0x7FFFFFFFu; if (first + b > tl.size()) return bad(who + ": instance range outside tlas_index"); for (size_t j = first; j < first + b; j++) if (tl[j] >= I) return bad(who + ": instance id out of range"); }
            else if (a >= tn || (b & NODE_INDEX_MASK) >= tn || a <= k || (b & NODE_INDEX_MASK) <= k) return bad(who + ": child index");
            for (int o = 0; o < 8; o++) if (esc[o] != NODE_TERM && (esc[o] >= tn || esc[o] <= k)) return bad(who + ": escape link");
        }
    }
    if (!NW) return MRT_OK;
    if (WNODE_STRIDE != 5) return MRT_OK;
    // ---- the TLAS slots (two-level scenes: a handful of nodes whose leaf children name instances): on the host
    const uint32_t tl_last = two ? std::min(NW, sc.tlas_wcap) : 0u;
    if (tl_last) {
        std::vector<float4> wn(5 * (size_t)tl_last);
        MRT_HIP(hipMemcpyAsync(wn.data(), wnodes, wn.size() * 16, hipMemcpyDeviceToHost, stream));
        MRT_HIP(hipStreamSynchronize(stream));
        for (uint32_t i = 0; i < tl_last; i++) {
            uint32_t w0[4], w1[4]; memcpy(w0, &wn[5 * (size_t)i], 16); memcpy(w1, &wn[5 * (size_t)i + 1], 16);
            const uint32_t imask = w0[3] >> 24, child_base = w1[0], tri_base = w1[1], meta[2] = {w1[2], w1[3]};
            const std::string who = "8-wide TLAS node " + std::to_string(i);
            const uint32_t ninner = (uint32_t)__builtin_popcount(imask);
            if (ninner) {
                if (child_base <= i) return bad(who + ": children do not come after their parent");
                if ((uint64_t)child_base + ninner > tl_last) return bad(who + ": internal children outside the TLAS slots");
            }
            for (int sl = 0; sl < 8; sl++) {
                const uint32_t m = (meta[sl >> 2] >> (8 * (sl & 3))) & 0xFFu, cnt = m >> 5, off = m & 31u;
                if (!cnt) continue;
                if ((imask >> sl) & 1u) return bad(who + ": slot " + std::to_string(sl) + " is both an internal and a leaf child");
                if (off + cnt > 32u) return bad(who + ": leaf range beyond the 32-bit triangle mask");
                if ((size_t)tri_base + off + cnt > wtl.size()) return bad(who + ": instance slot outside wtlas_index");
                for (uint32_t j = 0; j < cnt; j++) if (wtl[(size_t)tri_base + off + j] >= sc.num_inst) return bad(who + ": instance id out of range");
            }
        }
    }
    // ---- every other node (the flattened scene's tree, the BLASes): on the device, one thread per node, the first violation back in eight bytes
    const uint32_t first = tl_last, last = (two && tlas_only) ? tl_last : NW;
    if (last > first) {
        DevBuf<unsigned long long> d_err; MRT_HIP(d_err.alloc(1));
        MRT_HIP(hipMemsetAsync(d_err.p, 0xFF, 8, stream));
        hipLaunchKernelGGL(k_validate_wide, dim3((last - first + 255) / 256), dim3(256), 0, stream, wnodes, first, last, NW, sc.tlas_wcap, two ? 1u : 0u, (unsigned long long)packets, d_err.p);
        unsigned long long e = 0;
        MRT_HIP(hipMemcpyAsync(&e, d_err.p, 8, hipMemcpyDeviceToHost, stream));
        MRT_HIP(hipStreamSynchronize(stream));
        MRT_HIP(hipGetLastError());
        if (e != ~0ull) {
            static const char *what[] = {"?", "children do not come after their parent", "internal children outside wnodes", "a BLAS node points into the TLAS slots", "a slot is both an internal and a leaf child",
                                         "leaf range beyond the 32-bit triangle mask", "triangle packets outside wpackets"};
            return bad("8-wide node " + std::to_string((unsigned long long)(e >> 8)) + ": " + what[std::min<unsigned long long>(e & 0xFF, 6)]);
        }
    }
    return MRT_OK;
}

// instance rows + TLAS from the current transforms; BLAS data (out.bnodes, tri_shade, normals ...) is left alone
int update_tlas(const std::vector<HostMesh> &meshes, hipStream_t stream, DeviceScene &out) {
    const auto t0 = std::chrono::steady_clock::now();
    const size_t I = meshes.size();
    if (out.h_inst.size() != I) { set_error("update_tlas: the scene's instance list changed; commit rebuilds it"); return MRT_ERR_STATE; }
    std::vector<Box> boxes(I);
    std::vector<uint32_t> live;
    std::vector<float4> h_cols(std::max<size_t>(I * 4, 4));
    std::vector<float4> h_box(4 * std::max<size_t>(I, 1), make_float4(0, 0, 0, 0));      // + the instance's world box (padded, as the TLAS has it): what the flat TLAS pass queues pairs on      // per instance, the box of its BLAS in OBJECT space (the flat TLAS pass tests the transformed ray against it); lo > hi: no ray enters
    for (size_t i = 0; i < I; i++) {
        InstanceDev &d = out.h_inst[i];
        const float *xf = meshes[i].xf;
        for (int c = 0; c < 4; c++) h_cols[i * 4 + c] = make_float4(xf[c * 4 + 0], xf[c * 4 + 1], xf[c * 4 + 2], 0.0f);
        float rows[3][4];
        const bool ok = invert_affine(xf, rows);
        for (int r = 0; r < 3; r++) d.w2o[r] = ok ? make_float4(rows[r][0], rows[r][1], rows[r][2], rows[r][3]) : make_float4(0, 0, 0, 0);
        if (ok && d.ntri > 0) {
            boxes[i] = instance_box(xf, &out.blas_lo[3 * (size_t)d.blas], &out.blas_hi[3 * (size_t)d.blas]); live.push_back((uint32_t)i);
            h_box[4 * i] = make_float4(out.blas_lo[3 * (size_t)d.blas], out.blas_lo[3 * (size_t)d.blas + 1], out.blas_lo[3 * (size_t)d.blas + 2], 0);
            h_box[4 * i + 1] = make_float4(out.blas_hi[3 * (size_t)d.blas], out.blas_hi[3 * (size_t)d.blas + 1], out.blas_hi[3 * (size_t)d.blas + 2], 0);
            h_box[4 * i + 2] = make_float4(boxes[i].lo[0], boxes[i].lo[1], boxes[i].lo[2], 0); h_box[4 * i + 3] = make_float4(boxes[i].hi[0], boxes[i].hi[1], boxes[i].hi[2], 0);
        } else { h_box[4 * i] = make_float4(1, 1, 1, 0); h_box[4 * i + 1] = make_float4(-1, -1, -1, 0); h_box[4 * i + 2] = make_float4(1, 1, 1, 0); h_box[4 * i + 3] = make_float4(-1, -1, -1, 0); }
    }
    TlasBuilder tb(boxes);
    tb.order = live;
    if (!live.empty()) tb.build(0, (uint32_t)live.size(), NODE_TERM, 1);
    MRT_HIP(out.inst_box.alloc(h_box.size()));
    MRT_HIP(hipMemcpyAsync(out.inst_box.p, h_box.data(), h_box.size() * 16, hipMemcpyHostToDevice, stream));
    MRT_HIP(out.inst.alloc(std::max<size_t>(I, 1)));
    MRT_HIP(out.tlas_index.alloc(std::max<size_t>(tb.order.size(), 1)));
    MRT_HIP(out.nodes.alloc(std::max<size_t>(tb.nodes.size(), 8))); out.packets_offset = std::max<size_t>(tb.nodes.size(), 8);
    MRT_HIP(out.inst_cols.alloc(h_cols.size()));
    if (I) MRT_HIP(hipMemcpyAsync(out.inst.p, out.h_inst.data(), I * sizeof(InstanceDev), hipMemcpyHostToDevice, stream));
    if (!tb.order.empty()) MRT_HIP(hipMemcpyAsync(out.tlas_index.p, tb.order.data(), tb.order.size() * 4, hipMemcpyHostToDevice, stream));
    if (!tb.nodes.empty()) MRT_HIP(hipMemcpyAsync(out.nodes.p, tb.nodes.data(), tb.nodes.size() * 16, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(out.inst_cols.p, h_cols.data(), h_cols.size() * 16, hipMemcpyHostToDevice, stream));
    // 8-wide TLAS into the slots reserved in front of the BLASes (only when every BLAS has the 8-wide layout)
    WideTlasBuilder wb(boxes);
    out.num_wnodes = 0; out.wide_depth = 0;
    if (out.blas_wdepth > 0 && !live.empty()) {
        wb.build(live);
        const size_t wn = wb.nodes.size() / 5;
        if (wn <= out.tlas_wcap && wb.depth + 1 + out.blas_wdepth <= WIDE_STACK_TWO_LEVEL && WNODE_STRIDE == 5) {
            MRT_HIP(out.wtlas_index.alloc(std::max<size_t>(wb.order.size(), 1)));
            MRT_HIP(hipMemcpyAsync(out.wtlas_index.p, wb.order.data(), wb.order.size() * 4, hipMemcpyHostToDevice, stream));
            MRT_HIP(hipMemcpyAsync(out.wnodes.p, wb.nodes.data(), wb.nodes.size() * 16, hipMemcpyHostToDevice, stream));
            out.num_wnodes = (uint32_t)(out.wnodes.n / WNODE_STRIDE);
            out.wide_depth = wb.depth + 1 + out.blas_wdepth;
        }
    }
    MRT_HIP(hipStreamSynchronize(stream));
    out.num_inst = (uint32_t)I;
    out.stats.bvh_nodes = tb.nodes.size() / 4;          // TLAS nodes; the BLAS nodes are counted in scene_bytes
    out.rope_nodes = (uint32_t)(tb.nodes.size() / 4);
    out.stats.max_depth = tb.depth;
    out.tlas_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (out.validate) { if (int rc = validate_layout(out, stream, out.validated_blas)) return rc; out.validated_blas = true; }
    return MRT_OK;
}

int build_two_level(const std::vector<HostMesh> &meshes, const BuildOptions &opt_in, hipStream_t stream, DeviceScene &out) {
    const size_t I = meshes.size();
    BuildOptions opt = opt_in; opt.instancing = 0; opt.rope = 1;      // the BLASes keep both layouts: the stackless two-level walk serves the query API and is the fallback of the render kernels
    opt.presplit = 0.0f;                                                // one packet per triangle in a BLAS: the instance rows address packets by triangle count (ntri)                      // a BLAS is a flat scene of one mesh: rope layout (queries, A/B path) + 8-wide layout (render kernels)
    // distinct geometries, in order of first use
    std::vector<int> blas_of(I, -1); std::vector<size_t> blas_src;
    std::map<size_t, int> seen;
    size_t T_total = 0, V_total = 0; int max_sub = 1;
    for (size_t i = 0; i < I; i++) {
        const size_t src = meshes[i].source >= 0 ? (size_t)meshes[i].source : i;
        auto it = seen.find(src);
        if (it == seen.end()) { it = seen.emplace(src, (int)blas_src.size()).first; blas_src.push_back(src); }
        blas_of[i] = it->second;
        const HostMesh &g = meshes[src];
        max_sub = std::max<int>(max_sub, (int)g.sub_indices.size());
        for (auto &s : g.sub_indices) T_total += s.size() / 3;
    }
    if (I >= 65536 || max_sub >= 65536 || T_total >= (1ull << 32)) { set_error("scene too large (limits: 65535 instances / submeshes, 2^32 instanced triangles)"); return MRT_ERR_UNSUPPORTED; }
    const size_t B = blas_src.size();
    static const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    // ---- BLASes: each built into a scratch scene, surviving nodes + packets copied into the shared arrays
    std::vector<DeviceScene> blas(B);
    std::vector<uint32_t> node_base(B), packet_base(B), ts_base(B), vbase(B), ntri(B), wnode_base(B);
    size_t nodes_total = 0, packets_total = 0, ts_total = 0;
    const size_t tlas_wcap = std::max<size_t>(I, 1);                      // an 8-wide TLAS over I single-instance leaves has at most max(1, I - 1) nodes
    size_t wnodes_total = tlas_wcap; bool all_wide = B > 0; int blas_wdepth = 0;
    float build_ms = 0; double sah = 0, wcost = 0; uint64_t leaves = 0;
    out.blas_lo.assign(3 * B, 0.0f); out.blas_hi.assign(3 * B, 0.0f);
    out.blas_ranges.assign(B, BlasRange{});
    for (size_t b = 0; b < B; b++) {
        const HostMesh &g = meshes[blas_src[b]];
        std::vector<MeshRef> one{MeshRef{&g, identity}};
        int rc = build_flat(one, opt, stream, blas[b], &out.stage); if (rc) return rc;
        node_base[b] = (uint32_t)nodes_total; packet_base[b] = (uint32_t)packets_total; ts_base[b] = (uint32_t)ts_total; vbase[b] = (uint32_t)V_total;
        ntri[b] = (uint32_t)blas[b].stats.triangles;
        wnode_base[b] = (uint32_t)wnodes_total; wnodes_total += blas[b].num_wnodes;
        if (blas[b].stats.triangles > 0) { all_wide = all_wide && blas[b].num_wnodes > 0; blas_wdepth = std::max(blas_wdepth, blas[b].wide_depth); }
        nodes_total += blas[b].rope_nodes; packets_total += blas[b].stats.triangles; ts_total += blas[b].stats.triangles; V_total += g.positions.size() / 3;
        build_ms += blas[b].stats.build_ms; sah += blas[b].stats.sah_cost; wcost += blas[b].stats.wide_cost; leaves += blas[b].stats.bvh_leaves;
        {   // where this BLAS will lie in the shared arrays: what a refit of it needs (refit_two_level)
            BlasRange &r = out.blas_ranges[b];
            r.src_mesh = (uint32_t)blas_src[b]; r.wnode_base = wnode_base[b]; r.wnodes = blas[b].num_wnodes; r.packet_base = packet_base[b]; r.ntri = ntri[b]; r.node_base = node_base[b]; r.rope_nodes = blas[b].rope_nodes;
            r.ts_base = ts_base[b]; r.vbase = vbase[b]; r.wide_levels = blas[b].wide_levels; r.wide_cost_built = r.wide_cost = blas[b].stats.wide_cost; r.sah_cost_built = blas[b].stats.sah_cost;
        }
        for (int k = 0; k < 3; k++) { out.blas_lo[3 * b + k] = blas[b].root_lo[k]; out.blas_hi[3 * b + k] = blas[b].root_hi[k]; }
    }
    if (int rc = layout_limits(packets_total, nodes_total)) return rc;      // the shared arrays obey the same 24-bit / 32-bit addressing as one flat tree
    MRT_HIP(out.bnodes.alloc(std::max<size_t>(4 * nodes_total + 3 * packets_total, 8)));
    out.bpackets_offset = 4 * nodes_total;
    MRT_HIP(out.tri_shade.alloc(std::max<size_t>(ts_total, 1)));
    MRT_HIP(out.normals.alloc(std::max<size_t>(V_total, 1)));
    for (size_t b = 0; b < B; b++) {
        const size_t nn = blas[b].rope_nodes, nt = blas[b].stats.triangles, nv = meshes[blas_src[b]].positions.size() / 3;
        if (nn) MRT_HIP(hipMemcpyAsync(out.bnodes.p + 4 * (size_t)node_base[b], blas[b].nodes.p, nn * 64, hipMemcpyDeviceToDevice, stream));
        if (nt) MRT_HIP(hipMemcpyAsync(out.bnodes.p + out.bpackets_offset + 3 * (size_t)packet_base[b], blas[b].nodes.p + blas[b].packets_offset, nt * 48, hipMemcpyDeviceToDevice, stream));
        if (nt) MRT_HIP(hipMemcpyAsync(out.tri_shade.p + ts_base[b], blas[b].tri_shade.p, nt * 16, hipMemcpyDeviceToDevice, stream));
        if (nv) MRT_HIP(hipMemcpyAsync(out.normals.p + vbase[b], blas[b].normals.p, nv * 16, hipMemcpyDeviceToDevice, stream));
    }
    // 8-wide layout: [TLAS slots | BLAS 0 | BLAS 1 ...] with absolute indices; its packets are in the wide builder's own order (wpackets)
    all_wide = all_wide && wnodes_total < (1u << 24) && WNODE_STRIDE == 5;
    out.wnodes.release(); out.wpackets.release(); out.tri_packet.release(); out.tlas_wcap = 0; out.blas_wdepth = 0;
    if (all_wide) {
        MRT_HIP(out.wnodes.alloc(WNODE_STRIDE * wnodes_total)); MRT_HIP(out.wpackets.alloc(std::max<size_t>(WPK * packets_total, WPK)));
        MRT_HIP(hipMemsetAsync(out.wnodes.p, 0, out.wnodes.bytes(), stream));
        for (size_t b = 0; b < B; b++) {
            const size_t wn = blas[b].num_wnodes, nt = blas[b].stats.triangles;
            if (!wn) continue;
            MRT_HIP(hipMemcpyAsync(out.wnodes.p + WNODE_STRIDE * (size_t)wnode_base[b], blas[b].wnodes.p, wn * WNODE_STRIDE * 16, hipMemcpyDeviceToDevice, stream));
            MRT_HIP(hipMemcpyAsync(out.wpackets.p + WPK * (size_t)packet_base[b], blas[b].wpackets.p, nt * 16 * WPK, hipMemcpyDeviceToDevice, stream));
            hipLaunchKernelGGL(k_relocate_wide, dim3((uint32_t)((wn + 255) / 256)), dim3(256), 0, stream, out.wnodes.p + WNODE_STRIDE * (size_t)wnode_base[b], (uint32_t)wn, wnode_base[b], packet_base[b]);
        }
        // triangle -> packet, for the binned walk's results (renderer.hip k_shade<.., PAIRS>)
        MRT_HIP(out.tri_packet.alloc(std::max<size_t>(ts_total, 1)));
        MRT_HIP(hipMemsetAsync(out.tri_packet.p, 0, out.tri_packet.bytes(), stream));
        for (size_t b = 0; b < B; b++) {
            const uint32_t nt = (uint32_t)blas[b].stats.triangles;
            if (nt) hipLaunchKernelGGL(k_packet_map, dim3((nt + 255) / 256), dim3(256), 0, stream, (const float4 *)out.wpackets.p, packet_base[b], nt, ts_base[b], out.tri_packet.p);
        }
        MRT_HIP(hipGetLastError());
        out.tlas_wcap = (uint32_t)tlas_wcap; out.blas_wdepth = blas_wdepth;
    }
    MRT_HIP(hipStreamSynchronize(stream));
    blas.clear();
    // ---- per-instance tables: materials / first global triangle id per resource slot (Renderer.swift:128-151), instance rows
    std::vector<float4> h_base(std::max<size_t>(I * max_sub, 1), make_float4(0, 0, 0, 0));
    std::vector<uint32_t> h_gbase(std::max<size_t>(I * max_sub, 1), 0);
    std::vector<float4> h_mat(3 * std::max<size_t>(I * max_sub, 1), make_float4(0, 0, 0, 0));
    out.h_inst.assign(I, InstanceDev{});
    uint32_t gid = 0;
    for (size_t i = 0; i < I; i++) {
        const HostMesh &g = meshes[meshes[i].source >= 0 ? (size_t)meshes[i].source : i];
        InstanceDev &d = out.h_inst[i];
        const int b = blas_of[i];
        d.node_base = node_base[b]; d.packet_base = packet_base[b]; d.ts_base = ts_base[b]; d.vbase = vbase[b]; d.ntri = ntri[b]; d.blas = (uint32_t)b; d.gid_base = gid; d.wroot = wnode_base[b];
        uint32_t tb = gid;
        for (size_t s = 0; s < g.sub_indices.size(); s++) {
            h_base[i * max_sub + s] = make_float4(g.sub_materials[s].baseColor.x, g.sub_materials[s].baseColor.y, g.sub_materials[s].baseColor.z, 0.0f);
            pack_material(g.sub_materials[s], &h_mat[3 * (i * max_sub + s)]);
            h_gbase[i * max_sub + s] = tb;
            tb += (uint32_t)(g.sub_indices[s].size() / 3);
        }
        gid += ntri[b];
    }
    MRT_HIP(out.base_color.alloc(h_base.size())); MRT_HIP(out.geom_base.alloc(h_gbase.size()));
    MRT_HIP(hipMemcpyAsync(out.base_color.p, h_base.data(), h_base.size() * 16, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipMemcpyAsync(out.geom_base.p, h_gbase.data(), h_gbase.size() * 4, hipMemcpyHostToDevice, stream));
    MRT_HIP(out.materials.alloc(h_mat.size()));
    MRT_HIP(hipMemcpyAsync(out.materials.p, h_mat.data(), h_mat.size() * 16, hipMemcpyHostToDevice, stream));
    MRT_HIP(hipStreamSynchronize(stream));          // the host tables die at scope exit
    out.num_wnodes = 0; out.wide_depth = 0;
    out.stats = MRTSceneStats{};
    out.stats.triangles = T_total; out.stats.vertices = V_total; out.stats.instances = (int32_t)I; out.stats.max_submeshes = max_sub; out.stats.max_leaf_tris = opt.max_leaf;
    out.stats.bvh_leaves = leaves; out.stats.build_ms = build_ms; out.stats.sah_cost = B ? (float)(sah / (double)B) : 0.0f;
    out.stats.wide_cost = out.stats.wide_cost_built = (B && all_wide) ? (float)(wcost / (double)B) : 0.0f; out.sah_cost_built = out.stats.sah_cost; out.refits = 0; out.blas_all_wide = all_wide; out.stats.leaf_growth = 1.0f;
    out.stats.scene_bytes = (uint64_t)nodes_total * 64 + (uint64_t)packets_total * 48 + (all_wide ? (uint64_t)wnodes_total * 80 + (uint64_t)packets_total * 48 : 0) + (uint64_t)ts_total * 16 + (uint64_t)V_total * 16 + (uint64_t)I * (80 + 64 + (uint64_t)max_sub * 20);
    out.validated_blas = false;            // the BLAS part of wnodes is new: update_tlas checks all of it this time
    return update_tlas(meshes, stream, out);
}

// mrt_scene_update_mesh + commit on a two-level scene (Renderer.swift:193-213 is the instance-based API this stands for; Metal would refit the primitive acceleration structures of
// the changed meshes and rebuild the instance structure): every BLAS whose mesh changed is refitted IN PLACE in the shared arrays — both layouts: the 8-wide one the render kernels
// walk, the rope one the query API and the in-place fallbacks walk (bvh_build.hip refit_blas) — its instances share the result; then the TLAS is rebuilt from the new BLAS boxes
// (update_tlas).  MRT_ERR_UNSUPPORTED without a message: this scene cannot be refitted (no 8-wide layout, another mesh list) — the caller builds it.
int refit_two_level(const std::vector<HostMesh> &meshes, const BuildOptions &opt, hipStream_t stream, DeviceScene &out) {
    if (!out.blas_all_wide || out.num_inst != meshes.size() || out.blas_ranges.empty() || out.h_inst.size() != meshes.size()) return MRT_ERR_UNSUPPORTED;
    for (const BlasRange &r : out.blas_ranges) if (r.src_mesh >= meshes.size() || (meshes[r.src_mesh].dirty && (r.wnodes == 0 || r.wide_levels.empty() || meshes[r.src_mesh].positions.size() / 3 == 0))) return MRT_ERR_UNSUPPORTED;
    float ms_total = 0, growth_max = 1.0f; double sah = 0, wcost = 0;
    for (size_t b = 0; b < out.blas_ranges.size(); b++) {
        BlasRange &r = out.blas_ranges[b];
        if (meshes[r.src_mesh].dirty && r.ntri != 0) {
            float ms = 0, gr = 1.0f;
            if (int rc = refit_blas(meshes[r.src_mesh], r, stream, out, &out.blas_lo[3 * b], &out.blas_hi[3 * b], &ms, &gr)) return rc;
            r.leaf_growth *= gr;
            if (int rc = wide_tree_cost(out.wnodes.p, r.wnode_base, r.wnodes, r.wnode_base, opt.wide_cost_node, opt.wide_cost_tri, stream, &r.wide_cost)) return rc;
            ms_total += ms;
        }
        growth_max = std::max(growth_max, r.leaf_growth); wcost += r.wide_cost; sah += r.wide_cost_built > 0.0f ? r.sah_cost_built * (r.wide_cost / r.wide_cost_built) : r.sah_cost_built;
    }
    const MRTSceneStats before = out.stats;
    // scene option refit_max_cost_ratio: the refits have loosened the BLASes beyond that factor of their build-time cost — the caller builds the scene again
    if (opt.refit_max_cost_ratio > 0.0f && before.wide_cost_built > 0.0f && ((float)(wcost / (double)out.blas_ranges.size()) > opt.refit_max_cost_ratio * before.wide_cost_built || growth_max > opt.refit_max_cost_ratio)) return MRT_ERR_UNSUPPORTED;
    if (int rc = update_tlas(meshes, stream, out)) return rc;          // the instances' world boxes follow their BLAS's new root box
    const size_t B = out.blas_ranges.size();
    out.stats.build_ms = ms_total; out.stats.wide_cost = (float)(wcost / (double)B); out.stats.wide_cost_built = before.wide_cost_built; out.stats.sah_cost = (float)(sah / (double)B);
    out.refits++; out.stats.refits = out.refits; out.stats.leaf_growth = growth_max;
    return MRT_OK;
}

}  // namespace mrt
