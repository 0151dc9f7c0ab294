// traverse_instanced.h — two-level traversal (TLAS of instances over shared object-space BLASes; two_level.hip) for the two uses
// of `intersector<triangle_data, instancing>::intersect` (Raytracing.metal:244 closest, :367 any) when the scene was committed with
// instancing = 1.  The reference's intersector does exactly this walk inside Apple's driver: instance acceleration structure ->
// per-instance transform (MTLAccelerationStructureInstanceDescriptor.transformationMatrix, Renderer.swift:193-203) -> primitive structure.
//
// Both levels use the stackless rope layout of scene_device.h.  At an instance the ray is taken into object space with the instance's
// world->object rows, its direction NOT renormalised, so the parameter t of an object-space hit is the world-space distance and the
// closest-hit bound carries across instances unchanged.  Closest hit = global minimum t, ties to the lowest GLOBAL triangle id
// (instance-major, then geometry, then primitive: the numbering of the flattened scene), so the result does not depend on either tree.
#pragma once
#include "traverse.h"

namespace mrt {
namespace {

MRT_DEV f3 to_object_point(const InstanceDev &I, f3 p) {
    return mk3(__builtin_fmaf(I.w2o[0].z, p.z, __builtin_fmaf(I.w2o[0].y, p.y, I.w2o[0].x * p.x)) + I.w2o[0].w,
               __builtin_fmaf(I.w2o[1].z, p.z, __builtin_fmaf(I.w2o[1].y, p.y, I.w2o[1].x * p.x)) + I.w2o[1].w,
               __builtin_fmaf(I.w2o[2].z, p.z, __builtin_fmaf(I.w2o[2].y, p.y, I.w2o[2].x * p.x)) + I.w2o[2].w);
}
MRT_DEV f3 to_object_dir(const InstanceDev &I, f3 v) {
    return mk3(__builtin_fmaf(I.w2o[0].z, v.z, __builtin_fmaf(I.w2o[0].y, v.y, I.w2o[0].x * v.x)),
               __builtin_fmaf(I.w2o[1].z, v.z, __builtin_fmaf(I.w2o[1].y, v.y, I.w2o[1].x * v.x)),
               __builtin_fmaf(I.w2o[2].z, v.z, __builtin_fmaf(I.w2o[2].y, v.y, I.w2o[2].x * v.x)));
}

// slab test of one rope node (lo | a, hi | b) against a ray given as (1/d, -o/d); far side widened by ~4 ulp (Ize 2013)
MRT_DEV bool rope_box_hit(const float4 r0, const float4 r1, float ix, float iy, float iz, float nox, float noy, float noz, float tmin, float tmax) {
    const float tx0 = __builtin_fmaf(r0.x, ix, nox), tx1 = __builtin_fmaf(r1.x, ix, nox);
    const float ty0 = __builtin_fmaf(r0.y, iy, noy), ty1 = __builtin_fmaf(r1.y, iy, noy);
    const float tz0 = __builtin_fmaf(r0.z, iz, noz), tz1 = __builtin_fmaf(r1.z, iz, noz);
    const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
    const float tf = fminf(fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1)) * 1.0000005f, tmax);
    return tn <= tf;
}

// One flat loop, one kind of iteration (as traverse.h): every live lane does exactly one step per trip — a TLAS node, an instance entry,
// a BLAS node or a triangle — instead of a BLAS walk nested inside the TLAS walk, where the lanes of a wave that need the next TLAS step
// would wait for every other lane's whole BLAS walk to finish (measured on dragon x4: 2.0 Grays/s nested).
template <bool ANY, bool RUNTIME_ANY = false>
MRT_DEV bool traverse_instanced(const SceneView &s, f3 o, f3 d, float tmin, float tmax, TravHit &h, bool any_rt = false) {
    h.t = tmax; h.U = 0.0f; h.V = 0.0f; h.ad = 1.0f; h.gid = 0xFFFFFFFFu;
    if (s.num_nodes == 0 || s.num_inst == 0) return false;
    const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);
    const float nox = -(o.x * ix), noy = -(o.y * iy), noz = -(o.z * iz);
    const char *__restrict__ bbase = reinterpret_cast<const char *>(s.bnodes);
    const uint32_t bpk0 = (uint32_t)(reinterpret_cast<const char *>(s.bpackets) - bbase);
    uint32_t cur = 0;                 // next TLAS node
    uint32_t li = 0, li_end = 0;      // pending instances of the current TLAS leaf
    // inside an instance: object-space ray, the instance's bases, the BLAS cursor
    bool in_blas = false;
    f3 oo = o, dd = d; float jx = ix, jy = iy, jz = iz, mox = nox, moy = noy, moz = noz;
    uint32_t node_base = 0, packet_base = 0, gid_base = 0, oct = 0;
    uint32_t bc = NODE_TERM, tri = 0, tri_end = 0;
    for (;;) {
        const bool do_tri = tri < tri_end;
        if (!do_tri && in_blas && bc == NODE_TERM) in_blas = false;            // this instance is finished: back to the TLAS (no fetch needed, fall through)
        if (!do_tri && !in_blas) {
            if (li < li_end) {                                                  // enter the next instance of the TLAS leaf
                const InstanceDev I = s.inst[s.tlas_index[li++]];
                oo = to_object_point(I, o); dd = to_object_dir(I, d);
                jx = box_inv(dd.x); jy = box_inv(dd.y); jz = box_inv(dd.z);          // box tests only: v_rcp_f32 + one Newton step (traverse.h)
                mox = -(oo.x * jx); moy = -(oo.y * jy); moz = -(oo.z * jz);
                oct = (dd.x < 0.0f ? 1u : 0u) | (dd.y < 0.0f ? 2u : 0u) | (dd.z < 0.0f ? 4u : 0u);
                node_base = I.node_base; packet_base = I.packet_base; gid_base = I.gid_base;
                bc = 0; in_blas = true;
                continue;
            }
            if (cur == NODE_TERM) break;
            const float4 *__restrict__ nd = s.nodes + 4 * (size_t)cur;         // a TLAS node
            const float4 r0 = nd[0], r1 = nd[1], r2 = nd[2];
            const uint32_t a = __float_as_uint(r0.w), b = __float_as_uint(r1.w), esc = __float_as_uint(r2.x);    // the host TLAS has one escape link for all octants
            const bool hit = rope_box_hit(r0, r1, ix, iy, iz, nox, noy, noz, tmin, h.t);
            if (hit && (a & NODE_LEAF)) { li = a & 0x7FFFFFFFu; li_end = li + b; cur = esc; }
            else cur = hit ? a : esc;
            continue;
        }
        // a BLAS node or one triangle packet of the current instance: the same three 16-byte loads from one base (traverse.h)
        const uint32_t esc_off = 32u + ((oct >> 2) << 4), esc_lane = oct & 3u;
        const uint32_t off = do_tri ? bpk0 + (packet_base + tri) * 48u : (node_base + bc) << 6;
        const float4 q0 = *reinterpret_cast<const float4 *>(bbase + off);
        const float4 q1 = *reinterpret_cast<const float4 *>(bbase + off + 16u);
        const float4 q2 = *reinterpret_cast<const float4 *>(bbase + off + (do_tri ? 32u : esc_off));
        if (do_tri) {
            tri++;
            float t, U, V, ad;
            if (tri_test(q0, q1, q2, oo, dd, tmin, h.t, t, U, V, ad)) {
                if (ANY || (RUNTIME_ANY && any_rt)) return true;
                const uint32_t gid = gid_base + __float_as_uint(q0.w);
                if (t < h.t || gid < h.gid) { h.t = t; h.U = U; h.V = V; h.ad = ad; h.gid = gid; }
            }
        } else {
            const uint32_t a = __float_as_uint(q0.w), b = __float_as_uint(q1.w);
            const float escf = esc_lane == 0 ? q2.x : esc_lane == 1 ? q2.y : esc_lane == 2 ? q2.z : q2.w;
            const uint32_t esc = __float_as_uint(escf);
            const bool hit = rope_box_hit(q0, q1, jx, jy, jz, mox, moy, moz, tmin, h.t);
            const bool leaf = (a & NODE_LEAF) != 0;
            const uint32_t child = ((b >> (24 + oct)) & 1u) ? (b & NODE_INDEX_MASK) : a;
            bc = (hit && !leaf) ? child : esc;
            if (hit && leaf) { tri = a & 0x7FFFFFFFu; tri_end = tri + b; }
        }
    }
    return h.gid != 0xFFFFFFFFu;
}

// instance of a global triangle id: the last instance whose gid_base <= gid (the rows are in instance order, gid_base ascending)
MRT_DEV uint32_t instance_of_gid(const SceneView &s, uint32_t gid) {
    uint32_t lo = 0, hi = s.num_inst - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s.inst[mid].gid_base <= gid) lo = mid; else hi = mid - 1; }
    return lo;
}

}  // namespace
}  // namespace mrt
