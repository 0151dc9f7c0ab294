// api.cpp — implementation of include/mrt_abi.h.  Every entry point validates its arguments,
// catches C++ exceptions and returns a status code; nothing throws or aborts across the ABI.
// There is no CPU fallback: without a HIP device mrt_context_create fails (MRT_ERR_NO_DEVICE).
#include "api_types.h"
#include "abi_check.h"
#include "../../include/mrt_debug.h"
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <memory>
#include <new>

namespace mrt {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_err = buf;
    return e == hipErrorOutOfMemory ? MRT_ERR_OUT_OF_MEMORY : MRT_ERR_HIP;
}
}  // namespace mrt

static int bind_device(MRTContext ctx) { MRT_HIP(hipSetDevice(ctx->device)); return MRT_OK; }

extern "C" {

const char *mrt_last_error(void) { return mrt::g_err.c_str(); }
int mrt_abi_version(void) { return MRT_ABI_VERSION; }

// ---------------------------------------------------------------- context
int mrt_context_create(int device_id, MRTContext *out) {
    MRT_TRY
    REQUIRE(out, "mrt_context_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) { mrt::set_error("no HIP device available (GPU not available — there is no CPU fallback)"); return MRT_ERR_NO_DEVICE; }
    REQUIRE(device_id >= 0 && device_id < count, "mrt_context_create: device_id out of range");
    std::unique_ptr<MRTContext_> c(new MRTContext_());
    c->device = device_id;
    MRT_HIP(hipSetDevice(device_id));
    MRT_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
     // (without it the whole upload rides on `stream`)
    hipDeviceProp_t prop;
    MRT_HIP(hipGetDeviceProperties(&prop, device_id));
    snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
    *out = c.release();
    return MRT_OK;
    MRT_CATCH
}
int mrt_context_destroy(MRTContext ctx) {
    if (!ctx) return MRT_OK;
    // the reference's objects are reference-counted (ARC); a C handle is not, so the order is checked instead: a context outlives its scenes and renderers, a scene its renderers
    if (ctx->live != 0) { mrt::set_error("mrt_context_destroy: " + std::to_string(ctx->live) + " scene(s) / renderer(s) of this context are still alive: destroy them first"); return MRT_ERR_STATE; }
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
    return MRT_OK;
}
int mrt_context_set_stream(MRTContext ctx, void *hip_stream) {
    MRT_TRY
    REQUIRE(ctx, "mrt_context_set_stream: ctx is NULL");
    int rc = bind_device(ctx); if (rc) return rc;
    if (ctx->own_stream && ctx->stream) { MRT_HIP(hipStreamSynchronize(ctx->stream)); MRT_HIP(hipStreamDestroy(ctx->stream)); }
    if (hip_stream) { ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false; }
    else { MRT_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    return MRT_OK;
    MRT_CATCH
}
int mrt_context_device_name(MRTContext ctx, char *buf, size_t buflen) {
    REQUIRE(ctx && buf && buflen, "mrt_context_device_name: bad argument");
    snprintf(buf, buflen, "%s", ctx->name);
    return MRT_OK;
}

// ---------------------------------------------------------------- scene
int mrt_scene_create(MRTContext ctx, MRTScene *out) {
    MRT_TRY
    REQUIRE(ctx && out, "mrt_scene_create: bad argument");
    MRTScene s = new MRTScene_(); s->ctx = ctx; *out = s; ctx->live++;
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_destroy(MRTScene scene) {
    if (!scene) return MRT_OK;
    if (scene->renderers != 0) { mrt::set_error("mrt_scene_destroy: " + std::to_string(scene->renderers) + " renderer(s) still use this scene: destroy them first"); return MRT_ERR_STATE; }
    scene->ctx->live--;
    (void)hipSetDevice(scene->ctx->device);
    (void)hipStreamSynchronize(scene->ctx->stream);
    delete scene;
    return MRT_OK;
}
// What the device is handed must be finite: a NaN or an infinity in a position, a normal or a transform is refused here, with the call that brought it, instead of as a
// builder that "made no progress" or as NaN radiance (the reference never sees one: ModelIO's importer produces its meshes, Model.swift:16-21).
static bool all_finite(const float *p, size_t stride_bytes, size_t n, int comps) {
    uint32_t bad = 0;          // exponent field all ones = NaN or infinity; integer ops, no branch and no floating-point dependency chain in the loop
    for (size_t i = 0; i < n; i++) {
        const char *q = (const char *)p + i * stride_bytes;
        for (int k = 0; k < comps; k++) { uint32_t b; memcpy(&b, q + 4 * k, 4); bad |= ((b & 0x7F800000u) + 0x00800000u) >> 31; }
    }
    return bad == 0;
}
int mrt_scene_add_mesh(MRTScene scene, const float *positions, size_t pos_stride, const float *normals, size_t nrm_stride,
                       size_t nverts, const float *xf, int32_t *mesh_id) {
    MRT_TRY
    REQUIRE(scene && xf, "mrt_scene_add_mesh: bad argument");
    REQUIRE(nverts == 0 || (positions && normals), "mrt_scene_add_mesh: NULL vertex arrays");
    REQUIRE(pos_stride >= 12 && nrm_stride >= 12 && pos_stride % 4 == 0 && nrm_stride % 4 == 0, "mrt_scene_add_mesh: strides must be multiples of 4 and >= 12");
    REQUIRE(scene->meshes.size() < 65535, "mrt_scene_add_mesh: too many meshes");
    REQUIRE(all_finite(xf, 64, 1, 16), "mrt_scene_add_mesh: the transform holds a NaN or an infinity");
    REQUIRE(nverts == 0 || (all_finite(positions, pos_stride, nverts, 3) && all_finite(normals, nrm_stride, nverts, 3)), "mrt_scene_add_mesh: a position or a normal is NaN or infinite");
    mrt::HostMesh m;
    m.positions.resize(nverts * 3); m.normals.resize(nverts * 3);
    for (size_t i = 0; i < nverts; i++) {
        const float *p = (const float *)((const char *)positions + i * pos_stride);
        const float *n = (const float *)((const char *)normals + i * nrm_stride);
        for (int k = 0; k < 3; k++) { m.positions[i * 3 + k] = p[k]; m.normals[i * 3 + k] = n[k]; }
    }
    memcpy(m.xf, xf, 64);
    m.xf[3] = m.xf[7] = m.xf[11] = 0.0f; m.xf[15] = 1.0f;              // matrix4x4_drop_last_row (Utilities.swift:92-101)
    scene->meshes.push_back(std::move(m));
    scene->committed = false; scene->only_transforms_changed = false; scene->only_vertices_changed = false;
    if (mesh_id) *mesh_id = (int32_t)scene->meshes.size() - 1;
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_add_instance(MRTScene scene, int32_t source_mesh_id, const float *xf, int32_t *mesh_id) {
    MRT_TRY
    REQUIRE(scene && xf, "mrt_scene_add_instance: bad argument");
    REQUIRE(source_mesh_id >= 0 && (size_t)source_mesh_id < scene->meshes.size(), "mrt_scene_add_instance: source_mesh_id out of range");
    REQUIRE(scene->meshes.size() < 65535, "mrt_scene_add_instance: too many meshes");
    REQUIRE(all_finite(xf, 64, 1, 16), "mrt_scene_add_instance: the transform holds a NaN or an infinity");
    mrt::HostMesh m;
    const int src = scene->meshes[source_mesh_id].source;
    m.source = src >= 0 ? src : source_mesh_id;                         // instances of an instance share the original's geometry
    memcpy(m.xf, xf, 64);
    m.xf[3] = m.xf[7] = m.xf[11] = 0.0f; m.xf[15] = 1.0f;
    const int geometry_of = m.source;
    scene->meshes.push_back(std::move(m));
    scene->committed = false; scene->only_transforms_changed = false; scene->only_vertices_changed = false;
    if (mesh_id) *mesh_id = (int32_t)scene->meshes.size() - 1;
    if (!scene->opt.instancing) {           // a flattened scene stages a copy of the source's geometry per instance: the pinned area grows here (see mrt_mesh_add_submesh)
        const mrt::HostMesh &g = scene->meshes[geometry_of];
        size_t bytes = g.positions.size() / 3 * 28 + 4096;
        for (auto &ix : g.sub_indices) bytes += ix.size() * 4;
        scene->stage_need += bytes;
        if (scene->stage_need > scene->dev.stage.cap && bind_device(scene->ctx) == MRT_OK) { if (scene->dev.stage.reserve(std::max(scene->stage_need, 2 * scene->dev.stage.cap)) != hipSuccess) (void)hipGetLastError(); }
    }
    return MRT_OK;
    MRT_CATCH
}
int mrt_mesh_add_submesh(MRTScene scene, int32_t mesh_id, const uint32_t *indices, size_t ntris, const MRTMaterial *material, int32_t *geometry_id) {
    MRT_TRY
    REQUIRE(scene && material, "mrt_mesh_add_submesh: bad argument");
    REQUIRE(mesh_id >= 0 && (size_t)mesh_id < scene->meshes.size(), "mrt_mesh_add_submesh: mesh_id out of range");
    REQUIRE(ntris == 0 || indices, "mrt_mesh_add_submesh: NULL indices");
    mrt::HostMesh &m = scene->meshes[mesh_id];
    REQUIRE(m.source < 0, "mrt_mesh_add_submesh: the mesh is an instance (mrt_scene_add_instance) and shares its source's submeshes");
    REQUIRE(m.sub_indices.size() < 65535, "mrt_mesh_add_submesh: too many submeshes");
    size_t nv = m.positions.size() / 3;
    for (size_t i = 0; i < ntris * 3; i++) REQUIRE(indices[i] < nv, "mrt_mesh_add_submesh: vertex index out of range");
    m.sub_indices.emplace_back(indices, indices + ntris * 3);
    m.sub_materials.push_back(*material);
    scene->committed = false; scene->only_transforms_changed = false; scene->only_vertices_changed = false;
    if (geometry_id) *geometry_id = (int32_t)m.sub_indices.size() - 1;
    // the pinned staging area the commit uploads from grows HERE, as the geometry is handed over (by doubling: a handful of allocations whatever the mesh count), not inside
    // mrt_scene_commit: pinning 25 MB is ~1.1 ms, a fifth of DragonScene's commit.  Flattened scenes stage every instance's copy; a failure here is not an error (the commit retries).
    if (!scene->opt.instancing || m.source < 0) {
        scene->stage_need += ntris * 12 + (m.sub_indices.size() == 1 ? nv * 28 : 0) + 4096;
        if (scene->stage_need > scene->dev.stage.cap && bind_device(scene->ctx) == MRT_OK) { if (scene->dev.stage.reserve(std::max(scene->stage_need, 2 * scene->dev.stage.cap)) != hipSuccess) (void)hipGetLastError(); }
    }
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_add_obj(MRTScene scene, const char *obj_path, const float position[3], const float rotation[3], float scale, int32_t *mesh_id) {
    MRT_TRY
    REQUIRE(scene && obj_path && position && rotation, "mrt_scene_add_obj: bad argument");
    mrt::MeshData md;
    if (!mrt::load_obj(obj_path, md)) return MRT_ERR_IO;
    float xf[16];
    mrt::make_transform(position, rotation, scale, xf);
    int32_t id = -1;
    int rc = mrt_scene_add_mesh(scene, md.positions.data(), 12, md.normals.data(), 12, md.positions.size() / 3, xf, &id);
    if (rc) return rc;
    for (auto &s : md.submeshes) { rc = mrt_mesh_add_submesh(scene, id, s.indices.data(), s.indices.size() / 3, &s.material, nullptr); if (rc) return rc; }
    if (mesh_id) *mesh_id = id;
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_set_lights(MRTScene scene, const MRTLight *lights, int32_t count) {
    MRT_TRY
    REQUIRE(scene && count >= 0 && (count == 0 || lights), "mrt_scene_set_lights: bad argument");
    scene->lights.assign(lights, lights + count);
    if (scene->committed) { int rc = bind_device(scene->ctx); if (rc) return rc; return mrt::upload_lights(scene->lights.data(), count, scene->ctx->stream, scene->dev); }
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_set_option(MRTScene scene, const char *key, double value) {
    MRT_TRY
    REQUIRE(scene && key, "mrt_scene_set_option: bad argument");
    std::string k(key);
    if (k == "builder") { REQUIRE(value == 0 || value == 1 || value == 2, "builder must be 0 (Karras LBVH), 1 (PLOC) or 2 (binned SAH on the host)"); scene->opt.builder = (int)value; }
    else if (k == "max_leaf") { REQUIRE(value >= 1 && value <= 16, "max_leaf must be in [1,16]"); scene->opt.max_leaf = (int)value; }
    else if (k == "cost_trav") scene->opt.cost_trav = (float)value;
    else if (k == "cost_isect") scene->opt.cost_isect = (float)value;
    else if (k == "wide") scene->opt.wide = (int)value;
    else if (k == "rope") { REQUIRE(value == 0 || value == 1, "rope must be 0 (the rope layout only for scenes without the 8-wide one) or 1 (always)"); scene->opt.rope = (int)value; }
    else if (k == "presplit") { REQUIRE(value >= 0, "presplit must be >= 0 (0 = off; k: triangles longer than k x the mean extent are split into references)"); scene->opt.presplit = (float)value; }
    else if (k == "refit_fenced") { REQUIRE(value == 0 || value == 1, "refit_fenced must be 0 or 1"); scene->opt.refit_fenced = (int)value; }
    else if (k == "validate") { REQUIRE(value == 0 || value == 1, "validate must be 0 or 1"); scene->opt.validate = (int)value; }
    else if (k == "wide_collapse") { REQUIRE(value == 0 || value == 1, "wide_collapse must be 0 (greedy) or 1 (SAH-optimal)"); scene->opt.wide_collapse = (int)value; }
    else if (k == "wide_cost_node") { REQUIRE(value > 0, "wide_cost_node must be positive"); scene->opt.wide_cost_node = (float)value; }
    else if (k == "wide_cost_tri") { REQUIRE(value > 0, "wide_cost_tri must be positive"); scene->opt.wide_cost_tri = (float)value; }
    else if (k == "instancing") { REQUIRE(value == 0 || value == 1, "instancing must be 0 (flatten) or 1 (two-level: shared BLAS per mesh + TLAS)"); scene->opt.instancing = (int)value; }
    else if (k == "refit") scene->opt.refit = value != 0;
    else if (k == "refit_max_cost_ratio") { REQUIRE(value == 0 || value >= 1, "refit_max_cost_ratio must be 0 (off) or >= 1"); scene->opt.refit_max_cost_ratio = (float)value; }
    else if (k == "ploc_radius") { REQUIRE(value >= 1 && value <= 256, "ploc_radius must be in [1,256]"); scene->opt.ploc_radius = (int)value; }
    else { mrt::set_error("mrt_scene_set_option: unknown key " + k); return MRT_ERR_INVALID_ARGUMENT; }
    scene->committed = false; scene->only_transforms_changed = false; scene->only_vertices_changed = false;
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_commit(MRTScene scene) {
    MRT_TRY
    REQUIRE(scene, "mrt_scene_commit: scene is NULL");
    int rc = bind_device(scene->ctx); if (rc) return rc;
    if (scene->only_transforms_changed && scene->opt.instancing && scene->dev.num_inst == scene->meshes.size())
        rc = mrt::update_tlas(scene->meshes, scene->ctx->stream, scene->dev);         // instance rows + TLAS; the BLASes stay (the refit of an animated scene)
    else if (scene->only_vertices_changed && scene->opt.instancing && scene->opt.refit && (rc = mrt::refit_two_level(scene->meshes, scene->opt, scene->ctx->stream, scene->dev)) != MRT_ERR_UNSUPPORTED) {}      // the changed meshes' BLASes refitted in place + the TLAS
    else rc = mrt::build_scene(scene->meshes, scene->opt, scene->ctx->stream, scene->dev, scene->only_transforms_changed && !scene->opt.instancing, scene->only_vertices_changed && !scene->opt.instancing);
    if (rc) return rc;
    scene->only_transforms_changed = false; scene->only_vertices_changed = false;
    for (auto &m : scene->meshes) m.dirty = false;
    rc = mrt::upload_lights(scene->lights.data(), (int)scene->lights.size(), scene->ctx->stream, scene->dev); if (rc) return rc;
    scene->committed = true;
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_update_mesh(MRTScene scene, int32_t mesh_id, const float *positions, size_t pos_stride, const float *normals, size_t nrm_stride, size_t nverts) {
    MRT_TRY
    REQUIRE(scene && positions && normals, "mrt_scene_update_mesh: bad argument");
    REQUIRE(mesh_id >= 0 && (size_t)mesh_id < scene->meshes.size(), "mrt_scene_update_mesh: mesh_id out of range");
    REQUIRE(pos_stride >= 12 && nrm_stride >= 12 && pos_stride % 4 == 0 && nrm_stride % 4 == 0, "mrt_scene_update_mesh: strides must be multiples of 4 and >= 12");
    mrt::HostMesh &m = scene->meshes[(size_t)mesh_id];
    REQUIRE(m.source < 0, "mrt_scene_update_mesh: an instance has no vertices of its own (update its source mesh)");
    REQUIRE(nverts * 3 == m.positions.size(), "mrt_scene_update_mesh: the vertex count must stay the same (the topology is kept)");
    REQUIRE(all_finite(positions, pos_stride, nverts, 3) && all_finite(normals, nrm_stride, nverts, 3), "mrt_scene_update_mesh: a position or a normal is NaN or infinite (the mesh keeps what it had)");
    for (size_t i = 0; i < nverts; i++) {
        const float *p = (const float *)((const char *)positions + i * pos_stride);
        const float *n = (const float *)((const char *)normals + i * nrm_stride);
        for (int k = 0; k < 3; k++) { m.positions[i * 3 + k] = p[k]; m.normals[i * 3 + k] = n[k]; }
    }
    // a committed scene in which nothing else changes until the next commit keeps its tree: that commit refits (flattened scenes with the 8-wide layout; others build again)
    m.dirty = true;
    if (scene->committed) scene->only_vertices_changed = true;                         // the first change since the commit
    else if (!scene->only_vertices_changed) scene->only_transforms_changed = false;    // something else changed already (a transform, a mesh, an option): the next commit builds
    scene->committed = false;
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_set_instance_transform(MRTScene scene, int32_t mesh_id, const float *xf) {
    MRT_TRY
    REQUIRE(scene && xf, "mrt_scene_set_instance_transform: bad argument");
    REQUIRE(mesh_id >= 0 && (size_t)mesh_id < scene->meshes.size(), "mrt_scene_set_instance_transform: mesh_id out of range");
    REQUIRE(all_finite(xf, 64, 1, 16), "mrt_scene_set_instance_transform: the transform holds a NaN or an infinity (the instance keeps what it had)");
    float *m = scene->meshes[mesh_id].xf;
    memcpy(m, xf, 64);
    m[3] = m[7] = m[11] = 0.0f; m[15] = 1.0f;
    // flattened scene: the world-space BVH is rebuilt by the next mrt_scene_commit (22 ms for 885 K triangles);
    // two-level scene: the next commit rewrites the instance rows and rebuilds the TLAS only
    if (scene->committed) scene->only_transforms_changed = true;
    if (scene->only_vertices_changed) { scene->only_vertices_changed = false; scene->only_transforms_changed = false; }      // vertices AND transforms changed: a full build
    scene->committed = false;
    return MRT_OK;
    MRT_CATCH
}
int mrt_scene_stats(MRTScene scene, MRTSceneStats *out) {
    REQUIRE(scene && out, "mrt_scene_stats: bad argument");
    if (!scene->committed) { mrt::set_error("mrt_scene_stats: scene not committed"); return MRT_ERR_STATE; }
    *out = scene->dev.stats;
    out->wide_layout = scene->dev.num_wnodes > 0 ? 1 : 0; out->wide_depth = scene->dev.wide_depth;
    return MRT_OK;
}
int mrt_scene_instance_transform(MRTScene scene, int32_t mesh_id, float out[12]) {
    REQUIRE(scene && out, "mrt_scene_instance_transform: bad argument");
    REQUIRE(mesh_id >= 0 && (size_t)mesh_id < scene->meshes.size(), "mrt_scene_instance_transform: mesh_id out of range");
    const float *m = scene->meshes[mesh_id].xf;
    for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) out[c * 3 + r] = m[c * 4 + r];
    return MRT_OK;
}
int mrt_scene_intersect_closest(MRTScene scene, const MRTRay *rays, size_t n, MRTIntersection *out) {
    MRT_TRY
    REQUIRE(scene && (n == 0 || (rays && out)), "mrt_scene_intersect_closest: bad argument");
    if (!scene->committed) { mrt::set_error("mrt_scene_intersect_closest: scene not committed"); return MRT_ERR_STATE; }
    int rc = bind_device(scene->ctx); if (rc) return rc;
    return mrt::query_closest(scene->dev, scene->ctx->stream, rays, n, out);
    MRT_CATCH
}
int mrt_scene_intersect_any(MRTScene scene, const MRTRay *rays, size_t n, int32_t *occluded) {
    MRT_TRY
    REQUIRE(scene && (n == 0 || (rays && occluded)), "mrt_scene_intersect_any: bad argument");
    if (!scene->committed) { mrt::set_error("mrt_scene_intersect_any: scene not committed"); return MRT_ERR_STATE; }
    int rc = bind_device(scene->ctx); if (rc) return rc;
    return mrt::query_any(scene->dev, scene->ctx->stream, rays, n, occluded);
    MRT_CATCH
}

int mrt_debug_traversal_stats(MRTScene scene, const MRTRay *rays, size_t n, int32_t any_hit, uint32_t *out4) {
    MRT_TRY
    REQUIRE(scene && (n == 0 || (rays && out4)), "mrt_debug_traversal_stats: bad argument");
    if (!scene->committed) { mrt::set_error("mrt_debug_traversal_stats: scene not committed"); return MRT_ERR_STATE; }
    if (scene->dev.num_inst) { mrt::set_error("mrt_debug_traversal_stats: not available for two-level scenes"); return MRT_ERR_UNSUPPORTED; }
    int rc = bind_device(scene->ctx); if (rc) return rc;
    return mrt::query_stats(scene->dev, scene->ctx->stream, rays, n, any_hit, out4);
    MRT_CATCH
}

int mrt_debug_stream_stats(MRTScene scene, const MRTRay *rays, size_t n, int32_t any_hit, uint32_t per_wave, uint32_t *out8, size_t nwaves) {
    MRT_TRY
    REQUIRE(scene && rays && out8 && per_wave >= 64 && nwaves * (size_t)per_wave >= n, "mrt_debug_stream_stats: bad argument");
    if (!scene->committed) { mrt::set_error("mrt_debug_stream_stats: scene not committed"); return MRT_ERR_STATE; }
    int rc = bind_device(scene->ctx); if (rc) return rc;
    return mrt::query_stream_stats(scene->dev, scene->ctx->stream, rays, n, any_hit, per_wave, out8, nwaves);
    MRT_CATCH
}

int mrt_debug_intersect_stream(MRTScene scene, const MRTRay *rays, size_t n, int32_t any_hit, MRTIntersection *out) {
    MRT_TRY
    REQUIRE(scene && rays && out, "mrt_debug_intersect_stream: bad argument");
    if (!scene->committed) { mrt::set_error("mrt_debug_intersect_stream: scene not committed"); return MRT_ERR_STATE; }
    int rc = bind_device(scene->ctx); if (rc) return rc;
    return mrt::query_stream(scene->dev, scene->ctx->stream, rays, n, any_hit ? 1 : 0, out);
    MRT_CATCH
}

#ifdef MRT_WAVE_TIMES
extern "C++" { namespace mrt { int read_wave_times(unsigned long long *out); int read_wave_iters(uint32_t *out); int read_drain_probe(unsigned long long *out18, int reset); } }
extern "C" int mrt_debug_drain_probe(unsigned long long *out18, int reset) { return mrt::read_drain_probe(out18, reset); }
extern "C" int mrt_debug_wave_times(unsigned long long *out16384) { return mrt::read_wave_times(out16384); }
extern "C" int mrt_debug_wave_iters(uint32_t *out32768) { return mrt::read_wave_iters(out32768); }
#endif

// ---------------------------------------------------------------- host geometry helpers
int mrt_obj_load(const char *obj_path, MRTMeshData *out) {
    MRT_TRY
    REQUIRE(obj_path && out, "mrt_obj_load: bad argument");
    std::unique_ptr<MRTMeshData_> m(new MRTMeshData_());
    if (!mrt::load_obj(obj_path, m->m)) return MRT_ERR_IO;
    *out = m.release();
    return MRT_OK;
    MRT_CATCH
}
int mrt_dragon_proxy(MRTMeshData *out) {
    MRT_TRY
    REQUIRE(out, "mrt_dragon_proxy: out is NULL");
    std::unique_ptr<MRTMeshData_> m(new MRTMeshData_());
    mrt::make_dragon_proxy(m->m);
    *out = m.release();
    return MRT_OK;
    MRT_CATCH
}
int mrt_dragon_proxy_irregular(MRTMeshData *out) {
    MRT_TRY
    REQUIRE(out, "mrt_dragon_proxy_irregular: out is NULL");
    std::unique_ptr<MRTMeshData_> m(new MRTMeshData_());
    mrt::make_dragon_proxy_irregular(m->m);
    *out = m.release();
    return MRT_OK;
    MRT_CATCH
}
int mrt_dragon_proxy_hostile(MRTMeshData *out) {
    MRT_TRY
    REQUIRE(out, "mrt_dragon_proxy_hostile: out is NULL");
    std::unique_ptr<MRTMeshData_> m(new MRTMeshData_());
    mrt::make_dragon_proxy_hostile(m->m);
    *out = m.release();
    return MRT_OK;
    MRT_CATCH
}
int mrt_bunny_proxy(MRTMeshData *out) {
    MRT_TRY
    REQUIRE(out, "mrt_bunny_proxy: out is NULL");
    std::unique_ptr<MRTMeshData_> m(new MRTMeshData_());
    mrt::make_bunny_proxy(m->m);
    *out = m.release();
    return MRT_OK;
    MRT_CATCH
}
int mrt_meshdata_free(MRTMeshData m) { delete m; return MRT_OK; }
int mrt_meshdata_counts(MRTMeshData m, size_t *nverts, int32_t *nsub) {
    REQUIRE(m, "mrt_meshdata_counts: NULL");
    if (nverts) *nverts = m->m.positions.size() / 3;
    if (nsub) *nsub = (int32_t)m->m.submeshes.size();
    return MRT_OK;
}
int mrt_meshdata_vertices(MRTMeshData m, float *positions, float *normals) {
    REQUIRE(m, "mrt_meshdata_vertices: NULL");
    if (positions) memcpy(positions, m->m.positions.data(), m->m.positions.size() * 4);
    if (normals) memcpy(normals, m->m.normals.data(), m->m.normals.size() * 4);
    return MRT_OK;
}
int mrt_meshdata_submesh(MRTMeshData m, int32_t sub, size_t *ntris, uint32_t *indices, MRTMaterial *material, char *name_buf, size_t name_buflen) {
    REQUIRE(m, "mrt_meshdata_submesh: NULL");
    REQUIRE(sub >= 0 && (size_t)sub < m->m.submeshes.size(), "mrt_meshdata_submesh: submesh out of range");
    const mrt::Submesh &s = m->m.submeshes[sub];
    if (ntris) *ntris = s.indices.size() / 3;
    if (indices) memcpy(indices, s.indices.data(), s.indices.size() * 4);
    if (material) *material = s.material;
    if (name_buf && name_buflen) snprintf(name_buf, name_buflen, "%s", s.name.c_str());
    return MRT_OK;
}
int mrt_make_transform(const float position[3], const float rotation[3], float scale, float out16[16]) {
    REQUIRE(position && rotation && out16, "mrt_make_transform: bad argument");
    mrt::make_transform(position, rotation, scale, out16);
    return MRT_OK;
}
int mrt_default_camera(int32_t width, int32_t height, MRTCamera *out) {
    REQUIRE(out && width > 0 && height > 0, "mrt_default_camera: bad argument");
    mrt::default_camera(width, height, out);
    return MRT_OK;
}

// ---------------------------------------------------------------- renderer
int mrt_renderer_create(MRTContext ctx, MRTScene scene, int32_t width, int32_t height, uint32_t seed, int32_t max_bounces, MRTRenderer *out) {
    MRT_TRY
    REQUIRE(ctx && scene && out, "mrt_renderer_create: bad argument");
    REQUIRE(width > 0 && height > 0 && (int64_t)width * height < (1ll << 30), "mrt_renderer_create: bad size");
    REQUIRE(max_bounces >= 1 && max_bounces <= 19, "mrt_renderer_create: max_bounces must be in [1,19] (Halton prime table holds 100 primes)");
    REQUIRE(scene->ctx == ctx, "mrt_renderer_create: scene belongs to another context");
    if (!scene->committed) { mrt::set_error("mrt_renderer_create: scene not committed"); return MRT_ERR_STATE; }
    int rc = bind_device(ctx); if (rc) return rc;
    std::unique_ptr<MRTRenderer_> r(new MRTRenderer_());
    r->ctx = ctx; r->scene = scene;
    rc = r->r.init(ctx->stream, &scene->dev, width, height, seed, max_bounces); if (rc) return rc;
    *out = r.release(); scene->renderers++; ctx->live++;
    return MRT_OK;
    MRT_CATCH
}
int mrt_renderer_destroy(MRTRenderer r) {
    if (!r) return MRT_OK;
    r->scene->renderers--; r->ctx->live--;
    (void)hipSetDevice(r->ctx->device);
    (void)hipStreamSynchronize(r->ctx->stream);
    delete r;
    return MRT_OK;
}
#define RENDERER_PROLOGUE(name)                                     \
    REQUIRE(r, name ": renderer is NULL");                          \
    { int rc_ = bind_device(r->ctx); if (rc_) return rc_; }         \
    r->r.stream = r->ctx->stream;

int mrt_renderer_resize(MRTRenderer r, int32_t width, int32_t height) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_resize")
    REQUIRE(width > 0 && height > 0 && (int64_t)width * height < (1ll << 30), "mrt_renderer_resize: bad size");
    MRT_HIP(hipStreamSynchronize(r->r.stream));
    return r->r.resize(width, height);
    MRT_CATCH
}
int mrt_renderer_set_camera(MRTRenderer r, const MRTCamera *camera) {
    REQUIRE(r && camera, "mrt_renderer_set_camera: bad argument");
    r->r.camera = *camera;
    return MRT_OK;
}
// updateUniforms (Renderer.swift:216-229) as ONE call: size, frame index, light count and camera.  blocksWide is derived (the reference
// computes it from the size, :222, and its kernel never reads it).
int mrt_renderer_set_uniforms(MRTRenderer r, const MRTUniforms *u) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_set_uniforms")
    REQUIRE(u, "mrt_renderer_set_uniforms: uniforms is NULL");
    REQUIRE(u->width > 0 && u->height > 0 && (int64_t)u->width * u->height < (1ll << 30), "mrt_renderer_set_uniforms: bad size");
    REQUIRE(u->lightCount >= 1 && u->lightCount <= r->scene->dev.light_count, "mrt_renderer_set_uniforms: lightCount must be in [1, lights of the scene]");
    if (u->width != r->r.width || u->height != r->r.height) {          // mtkView(_:drawableSizeWillChange:): new targets, new seeds (Renderer.swift:353-356)
        MRT_HIP(hipStreamSynchronize(r->r.stream));
        int rc = r->r.resize(u->width, u->height); if (rc) return rc;
    }
    r->r.frame_index = u->frameIndex;
    r->r.light_count_limit = u->lightCount == r->scene->dev.light_count ? 0 : u->lightCount;
    r->r.camera = u->camera;
    return MRT_OK;
    MRT_CATCH
}
int mrt_renderer_get_uniforms(MRTRenderer r, MRTUniforms *u) {
    REQUIRE(r && u, "mrt_renderer_get_uniforms: bad argument");
    memset(u, 0, sizeof *u);
    u->width = r->r.width; u->height = r->r.height; u->blocksWide = (r->r.width + 7) / 8;
    u->frameIndex = r->r.frame_index;
    u->lightCount = r->r.light_count_limit > 0 ? std::min(r->r.light_count_limit, r->scene->dev.light_count) : r->scene->dev.light_count;
    u->camera = r->r.camera;
    return MRT_OK;
}
// commandBuffer.addCompletedHandler (Renderer.swift:285-287) as a poll: frames (since create / resize / reset_stats) whose accumulation
// has finished on the device.  Never blocks; mrt_renderer_wait is the blocking form.
int mrt_renderer_frames_completed(MRTRenderer r, uint64_t *frames) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_frames_completed")
    REQUIRE(frames, "mrt_renderer_frames_completed: NULL");
    return r->r.poll_completed(frames);
    MRT_CATCH
}
// The host's knobs: the reference's own (the bounce count, Raytracing.metal:237; maxFramesInFlight, Renderer.swift:33; the shard's sample offset) + three of this implementation.
int mrt_renderer_set_option(MRTRenderer r, const char *key, double value) {
    MRT_TRY
    REQUIRE(r && key, "mrt_renderer_set_option: bad argument");
    std::string k(key);
    if (k == "max_bounces") { REQUIRE(value >= 1 && value <= (r->r.materials ? 16 : 19), "max_bounces must be in [1,19] ([1,16] with materials = 1)"); r->r.max_bounces = (int)value; }
    else if (k == "frames_in_flight") { REQUIRE(value >= 1 && value <= mrt::MAX_FRAMES_IN_FLIGHT, "frames_in_flight must be in [1,16]"); r->r.frames_in_flight = (int)value; }
    else if (k == "sample_offset") { REQUIRE(value >= 0 && value < 4294967296.0, "sample_offset out of range"); r->r.sample_offset = (uint32_t)value; }
    else if (k == "frame_batch") { REQUIRE(value >= 0 && value <= mrt::MAX_FRAME_BATCH, "frame_batch must be in [1,32], or 0 for the default (by image size)"); r->r.frame_batch = (int)value; }
    else if (k == "megakernel") r->r.megakernel = value != 0;
    else if (k == "materials") { REQUIRE(value == 0 || (value == 1 && r->r.max_bounces <= 16), "materials must be 0 or 1 (and max_bounces <= 16: the lobe choice uses Halton dimension 2 + 5 * max_bounces + bounce < 100)"); r->r.materials = value != 0; }
    else { mrt::set_error("mrt_renderer_set_option: unknown key " + k + " (keys: max_bounces, frames_in_flight, sample_offset, frame_batch, megakernel, materials; the library's A/B switches are behind mrt_debug_renderer_set_option)"); return MRT_ERR_INVALID_ARGUMENT; }
    return MRT_OK;
    MRT_CATCH
}
int mrt_renderer_get_option(MRTRenderer r, const char *key, double *value) {
    MRT_TRY
    REQUIRE(r && key && value, "mrt_renderer_get_option: bad argument");
    std::string k(key);
    if (k == "max_bounces") *value = r->r.max_bounces;
    else if (k == "frames_in_flight") *value = r->r.frames_in_flight;
    else if (k == "sample_offset") *value = r->r.sample_offset;
    else if (k == "frame_batch") *value = r->r.batch_wanted();          // (what is in force: under the default, 8 at 1080p and above, up to 32 for smaller images and shards)
    else if (k == "megakernel") *value = r->r.megakernel ? 1 : 0;
    else if (k == "materials") *value = r->r.materials ? 1 : 0;
    else if (k == "lanes_used") *value = r->r.lanes_used;
    else if (k == "lane_bytes") *value = (double)r->r.lane_bytes();
    else { mrt::set_error("mrt_renderer_get_option: unknown key " + k); return MRT_ERR_INVALID_ARGUMENT; }
    return MRT_OK;
    MRT_CATCH
}
// The library's A/B switches and launch-shape parameters (tests, tools/, bench.py --opt): every setting renders the same image bit for bit.  Not part of the host contract:
// keys come and go with the experiments that need them.  Also accepts the public keys.
int mrt_debug_renderer_set_option(MRTRenderer r, const char *key, double value) {
    MRT_TRY
    REQUIRE(r && key, "mrt_debug_renderer_set_option: bad argument");
    std::string k(key);
    if (k == "persistent") { REQUIRE(value == 0 || value == 1 || value == 2, "persistent must be 0 (never), 1 (always) or 2 (by launch size)"); r->r.persistent = (int)value; }
    else if (k == "xcd_counters") { REQUIRE(value == 0 || value == 1, "xcd_counters must be 0 or 1"); r->r.xcd_counters = (int)value; }
    else if (k == "tile_groups") { REQUIRE(value >= 0 && value <= mrt::MAX_TILE_GROUPS && value == (int)value, "tile_groups must be 0 (by the draw), 1 (never) or 2..4"); r->r.tile_groups = (int)value; }
    else if (k == "shade_pack") { REQUIRE(value == 0 || value == 1, "shade_pack must be 0 or 1"); r->r.shade_pack = (int)value; }
    else if (k == "hit_lds") { REQUIRE(value == 0 || value == 1, "hit_lds must be 0 or 1"); r->r.hit_lds = (int)value; }
    else if (k == "persist_chunk") { REQUIRE(value >= 64 && value <= 65536 && ((int)value % 64) == 0, "persist_chunk must be a multiple of 64 in [64, 65536]"); r->r.persist_chunk = (int)value; }
    else if (k == "wave_slots") { REQUIRE(value >= 1 && value <= (1 << 20), "wave_slots must be in [1, 2^20]"); r->r.wave_slots = (int)value; r->r.wave_slots_user = true; }
    else if (k == "stream_stride") { REQUIRE(value >= 0 && value <= 2, "stream_stride must be 0 (contiguous ranges), 1 (round-robin batches) or 2 (round-robin for a shard's launches)"); r->r.stream_stride = (int)value; }
    else if (k == "halton_table") r->r.halton_table = value != 0;
    else if (k == "equal_passes") r->r.equal_passes = value != 0;
    else if (k == "frame_bundle") { REQUIRE(value == 0 || value == 1, "frame_bundle must be 0 (off) or 1 (the sub-frames of a slot side by side in a wave)"); r->r.frame_bundle = (int)value; }
    else if (k == "stream_even") { REQUIRE(value >= 0 && value <= 1600, "stream_even must be in [0,1600] (percent of the wave slots; 0 = off)"); r->r.stream_even = (int)value; }
    else if (k == "primary_hint") r->r.primary_hint = value != 0;
    else if (k == "throughput_chain") r->r.throughput_chain = value != 0;
    else if (k == "shadow_planes") r->r.shadow_planes = value != 0 ? 1 : 0;
    else if (k == "tail_accumulate") r->r.tail_accumulate = value != 0;
    else if (k == "fuse_primary") { REQUIRE(value == 0 || value == 1 || value == 2, "fuse_primary must be 0, 1 (not for one frame alone) or 2 (always)"); r->r.fuse_primary = (int)value; }
    else if (k == "wide_bounce") r->r.wide_bounce = value != 0;
    else if (k == "tl_pair_cap") { REQUIRE(value >= 0 && value <= 4294967295.0, "tl_pair_cap out of range"); r->r.tl_pair_cap = (int)std::min(value, 2147483647.0); }
    else if (k == "tl_pairs") { REQUIRE(value == 0 || value == 1 || value == 2, "tl_pairs must be 0 (two-level scenes walked in one loop), 1 (TLAS pass + BLAS pass over (ray, instance) pairs) or 2 (the same, the TLAS pass always as a walk of the 8-wide TLAS)"); r->r.tl_pairs = (int)value; }
    else if (k == "primary_wide") { REQUIRE(value == 0 || value == 1 || value == 2, "primary_wide must be 0 (rope walk inside shade(0)), 1 (own launch of the 8-wide stream kernel) or 2 (8-wide walk inside shade(0))"); r->r.primary_wide = (int)value; }
    else return mrt_renderer_set_option(r, key, value);
    return MRT_OK;
    MRT_CATCH
}
int mrt_debug_renderer_get_option(MRTRenderer r, const char *key, double *value) {
    MRT_TRY
    REQUIRE(r && key && value, "mrt_debug_renderer_get_option: bad argument");
    std::string k(key);
    if (k == "persistent") *value = r->r.persistent;
    else if (k == "persist_chunk") *value = r->r.persist_chunk;
    else if (k == "hit_lds") *value = r->r.hit_lds;
    else if (k == "shade_pack") *value = r->r.shade_pack;
    else if (k == "tile_groups") *value = r->r.tile_groups;
    else if (k == "groups_used") *value = r->r.groups_used;
    else if (k == "xcd_counters") *value = r->r.xcd_counters;
    else if (k == "wave_slots") *value = r->r.wave_slots;
    else if (k == "stream_stride") *value = r->r.stream_stride;
    else if (k == "halton_table") *value = r->r.halton_table;
    else if (k == "frame_bundle") *value = r->r.frame_bundle;
    else if (k == "stream_even") *value = r->r.stream_even;
    else if (k == "primary_hint") *value = r->r.primary_hint ? 1 : 0;
    else if (k == "throughput_chain") *value = r->r.throughput_chain ? 1 : 0;
    else if (k == "shadow_planes") *value = r->r.shadow_planes;
    else if (k == "tail_accumulate") *value = r->r.tail_accumulate ? 1 : 0;
    else if (k == "fuse_primary") *value = r->r.fuse_primary;
    else if (k == "wide_bounce") *value = r->r.wide_bounce ? 1 : 0;
    else if (k == "tl_pairs") *value = r->r.tl_pairs;
    else if (k == "tl_pair_cap") *value = r->r.tl_pair_cap;
    else if (k == "primary_wide") *value = r->r.primary_wide;
    else return mrt_renderer_get_option(r, key, value);
    return MRT_OK;
    MRT_CATCH
}
int mrt_renderer_set_shard(MRTRenderer r, int32_t rank, int32_t world) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_set_shard")
    return r->r.set_shard(rank, world);
    MRT_CATCH
}
int mrt_renderer_set_frame_index(MRTRenderer r, uint32_t fi) { REQUIRE(r, "mrt_renderer_set_frame_index: NULL"); r->r.frame_index = fi; return MRT_OK; }
int mrt_renderer_frame_index(MRTRenderer r, uint32_t *fi) { REQUIRE(r && fi, "mrt_renderer_frame_index: bad argument"); *fi = r->r.frame_index; return MRT_OK; }
int mrt_renderer_render(MRTRenderer r, int32_t n_frames) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_render")
    REQUIRE(n_frames >= 0, "mrt_renderer_render: n_frames < 0");
    if (!r->scene->committed) { mrt::set_error("mrt_renderer_render: scene was modified and not re-committed"); return MRT_ERR_STATE; }
    if (n_frames == 0) return MRT_OK;
    return r->r.render(n_frames);
    MRT_CATCH
}
int mrt_renderer_wait(MRTRenderer r) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_wait")
    return r->r.wait();
    MRT_CATCH
}
int mrt_renderer_read_accum(MRTRenderer r, float *rgba, size_t nbytes) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_read_accum")
    REQUIRE(rgba, "mrt_renderer_read_accum: NULL buffer");
    return r->r.read_accum(rgba, nbytes);
    MRT_CATCH
}
int mrt_renderer_copy_accum_to_device(MRTRenderer r, void *dptr, size_t nbytes) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_copy_accum_to_device")
    REQUIRE(dptr, "mrt_renderer_copy_accum_to_device: NULL pointer");
    return r->r.copy_accum_to_device(dptr, nbytes);
    MRT_CATCH
}
int mrt_renderer_write_accum_from_device(MRTRenderer r, const void *dptr, size_t nbytes) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_write_accum_from_device")
    REQUIRE(dptr, "mrt_renderer_write_accum_from_device: NULL pointer");
    return r->r.write_accum_from_device(dptr, nbytes);
    MRT_CATCH
}
int mrt_renderer_shard_tiles(MRTRenderer r, int32_t rank, int32_t world, uint64_t *tiles) {
    REQUIRE(r && tiles && world >= 1 && rank >= 0 && rank < world, "mrt_renderer_shard_tiles: bad argument");
    *tiles = mrt::shard_tiles(r->r.width, r->r.height, rank, world);
    return MRT_OK;
}
int mrt_renderer_pack_owned_tiles(MRTRenderer r, void *dptr, size_t nbytes) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_pack_owned_tiles")
    REQUIRE(dptr || nbytes == 0, "mrt_renderer_pack_owned_tiles: NULL pointer");
    return r->r.pack_owned_tiles(dptr, nbytes);
    MRT_CATCH
}
int mrt_renderer_unpack_tiles(MRTRenderer r, const void *dptr, size_t nbytes, int32_t rank, int32_t world) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_unpack_tiles")
    REQUIRE(dptr || nbytes == 0, "mrt_renderer_unpack_tiles: NULL pointer");
    return r->r.unpack_tiles(dptr, nbytes, rank, world);
    MRT_CATCH
}
int mrt_renderer_read_tonemapped_rgba8(MRTRenderer r, uint8_t *rgba, size_t nbytes) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_read_tonemapped_rgba8")
    REQUIRE(rgba, "mrt_renderer_read_tonemapped_rgba8: NULL buffer");
    return r->r.read_tonemapped(rgba, nbytes);
    MRT_CATCH
}
int mrt_renderer_stats(MRTRenderer r, MRTRenderStats *out) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_stats")
    REQUIRE(out, "mrt_renderer_stats: NULL");
    return r->r.stats(out);
    MRT_CATCH
}
int mrt_renderer_kernel_times(MRTRenderer r, MRTKernelTimes *out) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_kernel_times")
    REQUIRE(out, "mrt_renderer_kernel_times: NULL");
    int rc = r->r.wait(); if (rc) return rc;
    *out = r->r.kernel_times;
    return MRT_OK;
    MRT_CATCH
}
int mrt_renderer_reset_stats(MRTRenderer r) {
    MRT_TRY
    RENDERER_PROLOGUE("mrt_renderer_reset_stats")
    return r->r.reset_stats();
    MRT_CATCH
}

// ---------------------------------------------------------------- probes
int mrt_debug_halton(MRTContext ctx, const int32_t *i, const int32_t *d, size_t n, float *out) {
    MRT_TRY
    REQUIRE(ctx && (n == 0 || (i && d && out)), "mrt_debug_halton: bad argument");
    for (size_t k = 0; k < n; k++) REQUIRE(d[k] >= 0 && d[k] < 100, "mrt_debug_halton: 0 <= d < 100 required");
    int rc = bind_device(ctx); if (rc) return rc;
    return mrt::probe_halton(ctx->stream, i, d, n, out);
    MRT_CATCH
}
int mrt_debug_hemisphere(MRTContext ctx, const float *u2, const float *n3, size_t n, float *out3) {
    MRT_TRY
    REQUIRE(ctx && (n == 0 || (u2 && n3 && out3)), "mrt_debug_hemisphere: bad argument");
    int rc = bind_device(ctx); if (rc) return rc;
    return mrt::probe_hemisphere(ctx->stream, u2, n3, n, out3);
    MRT_CATCH
}
int mrt_debug_seeds(MRTContext ctx, uint32_t seed, int32_t width, int32_t height, uint32_t *out) {
    MRT_TRY
    REQUIRE(ctx && out && width > 0 && height > 0, "mrt_debug_seeds: bad argument");
    int rc = bind_device(ctx); if (rc) return rc;
    return mrt::probe_seeds(ctx->stream, seed, width, height, out);
    MRT_CATCH
}

int mrt_debug_wide_histogram(MRTScene scene, uint32_t *out12) {
    MRT_TRY
    REQUIRE(scene && out12, "mrt_debug_wide_histogram: bad argument");
    if (!scene->committed) { mrt::set_error("mrt_debug_wide_histogram: scene not committed"); return MRT_ERR_STATE; }
    int rc = bind_device(scene->ctx); if (rc) return rc;
    return mrt::wide_histogram(scene->dev, scene->ctx->stream, out12);
    MRT_CATCH
}
// the 8-wide nodes of a committed scene as they lie in device memory (tests: two builds of one scene must agree bit for bit)
int mrt_debug_read_wnodes(MRTScene scene, void *out, size_t nbytes, uint64_t *num_nodes) {
    MRT_TRY
    REQUIRE(scene && num_nodes, "mrt_debug_read_wnodes: bad argument");
    if (!scene->committed) { mrt::set_error("mrt_debug_read_wnodes: scene not committed"); return MRT_ERR_STATE; }
    *num_nodes = scene->dev.num_wnodes;
    const size_t have = (size_t)scene->dev.num_wnodes * mrt::WNODE_STRIDE * 16;
    if (out == nullptr) return MRT_OK;
    REQUIRE(nbytes == have, "mrt_debug_read_wnodes: nbytes must be num_nodes x 16 x the node stride (80 bytes per node)");
    int rc = bind_device(scene->ctx); if (rc) return rc;
    MRT_HIP(hipMemcpy(out, scene->dev.wnodes.p, have, hipMemcpyDeviceToHost));
    return MRT_OK;
    MRT_CATCH
}
// host wall time of the last commit of a flattened scene by phase (ms): staging, device allocations, topology, 8-wide emit, rope emit, validation
int mrt_debug_scene_refits(MRTScene scene, uint32_t *out) {
    REQUIRE(scene && out, "mrt_debug_scene_refits: bad argument");
    *out = scene->dev.refits;
    return MRT_OK;
}
int mrt_debug_commit_times(MRTScene scene, double *out6) {
    REQUIRE(scene && out6, "mrt_debug_commit_times: bad argument");
    for (int k = 0; k < 6; k++) out6[k] = scene->dev.commit_ms[k];
    return MRT_OK;
}
// the commit-time validator on demand, and a way for its test to break one word of one 8-wide node (tests/test_instancing.py)
int mrt_debug_validate(MRTScene scene) {
    MRT_TRY
    REQUIRE(scene, "mrt_debug_validate: scene is NULL");
    if (!scene->committed) { mrt::set_error("mrt_debug_validate: scene not committed"); return MRT_ERR_STATE; }
    int rc = bind_device(scene->ctx); if (rc) return rc;
    return mrt::validate_layout(scene->dev, scene->ctx->stream, false);
    MRT_CATCH
}
int mrt_debug_validate_patched(MRTScene scene, uint32_t node, uint32_t word, uint32_t value) {
    MRT_TRY
    REQUIRE(scene && scene->committed && node < scene->dev.num_wnodes && word < 4 * mrt::WNODE_STRIDE, "mrt_debug_validate_patched: bad argument");
    int rc = bind_device(scene->ctx); if (rc) return rc;
    mrt::DevBuf<float4> copy; MRT_HIP(copy.alloc(scene->dev.wnodes.n));
    MRT_HIP(hipMemcpy(copy.p, scene->dev.wnodes.p, scene->dev.wnodes.bytes(), hipMemcpyDeviceToDevice));
    MRT_HIP(hipMemcpy(reinterpret_cast<uint32_t *>(copy.p + (size_t)mrt::WNODE_STRIDE * node) + word, &value, 4, hipMemcpyHostToDevice));
    return mrt::validate_layout(scene->dev, scene->ctx->stream, false, copy.p);
    MRT_CATCH
}
#ifdef MRT_DIAGNOSTICS
int mrt_debug_poke_wnode(MRTScene scene, uint32_t node, uint32_t word, uint32_t value, uint32_t *old_value) {
    MRT_TRY
    REQUIRE(scene && scene->committed && node < scene->dev.num_wnodes && word < 4 * mrt::WNODE_STRIDE, "mrt_debug_poke_wnode: bad argument");
    int rc = bind_device(scene->ctx); if (rc) return rc;
    uint32_t *p = reinterpret_cast<uint32_t *>(scene->dev.wnodes.p + (size_t)mrt::WNODE_STRIDE * node) + word;
    if (old_value) MRT_HIP(hipMemcpy(old_value, p, 4, hipMemcpyDeviceToHost));
    MRT_HIP(hipMemcpy(p, &value, 4, hipMemcpyHostToDevice));
    return MRT_OK;
    MRT_CATCH
}
#endif
// builder = 2's host part on caller boxes (n x {lo.xyz, -} and {hi.xyz, -}): leaf order, left / right of the n - 1 internal nodes, parent of all 2n - 1 (no device needed)
int mrt_debug_host_sah(const float *lo4, const float *hi4, uint32_t n, uint32_t *order, uint32_t *left, uint32_t *right, uint32_t *parent) {
    MRT_TRY
    REQUIRE(lo4 && hi4 && n >= 1 && order && left && right && parent, "mrt_debug_host_sah: bad argument");
    std::vector<uint32_t> o, l, r, p;
    mrt::host_sah_topology(reinterpret_cast<const float4 *>(lo4), reinterpret_cast<const float4 *>(hi4), n, o, l, r, p);
    memcpy(order, o.data(), (size_t)n * 4); memcpy(parent, p.data(), (2 * (size_t)n - 1) * 4);
    if (n > 1) { memcpy(left, l.data(), (size_t)(n - 1) * 4); memcpy(right, r.data(), (size_t)(n - 1) * 4); }
    return MRT_OK;
    MRT_CATCH
}
int mrt_debug_layout_limits(uint64_t triangles, uint64_t nodes) {
    MRT_TRY
    return mrt::layout_limits(triangles, nodes);
    MRT_CATCH
}
int mrt_debug_calibrate(MRTContext ctx, size_t table_bytes, double *out3) {
    MRT_TRY
    REQUIRE(ctx && out3 && table_bytes >= 4096 && table_bytes <= ((size_t)1 << 34), "mrt_debug_calibrate: bad argument");
    int rc = bind_device(ctx); if (rc) return rc;
    return mrt::calibrate(ctx->stream, table_bytes, out3);
    MRT_CATCH
}

}  // extern "C"
