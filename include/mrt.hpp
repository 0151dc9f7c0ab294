// mrt.hpp — C++17 host-side mirror of the reference's Swift classes over the C ABI (mrt_abi.h).
//
// The reference's host side is Swift (Scene.swift, DragonScene.swift, Model.swift, Mesh.swift,
// SubMesh.swift, Renderer.swift); no Swift toolchain exists in this environment, so the host layer above
// the C ABI is C++ (header only), with the same type names, initialiser arguments and properties:
//
//   Swift                                               here
//   Model(name:position:rotation:scale:on:)             mrt::Model(name, position, rotation, scale)
//   Mesh.transform / .submeshes                         mrt::Mesh::transform / submeshes
//   Submesh.material                                    mrt::Submesh::material
//   Scene.models / .camera / .lights                    mrt::Scene::models / camera / lights
//   Light.areaLight/sunLight/pointLight/spotLight       mrt::Light::areaLight/...
//   DragonScene(size:device:)                           mrt::DragonScene(width, height)
//   Renderer(metalView:) / draw(in:) / frameIndex       mrt::Renderer(width, height, scene) / draw() / frameIndex()
//
// Errors: the Swift code traps (fatalError, try!); here every failing ABI call throws mrt::Error.
#pragma once
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "mrt_abi.h"

namespace mrt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("mrt error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) { if (rc != MRT_OK) throw Error(rc, mrt_last_error()); }

using Camera = MRTCamera;
using Material = MRTMaterial;

inline MRTFloat3 f3(float x, float y, float z) { return MRTFloat3{x, y, z, 0.0f}; }

struct Light : MRTLight {                                            // Scene.swift:70-107
    Light() { std::memset(static_cast<MRTLight *>(this), 0, sizeof(MRTLight)); }
    static Light areaLight(MRTFloat3 position, MRTFloat3 forward, MRTFloat3 right, MRTFloat3 up, MRTFloat3 color) {
        Light l; l.type = MRTLightTypeAreaLight; l.position = position; l.forward = forward; l.right = right; l.up = up; l.color = color; return l;
    }
    static Light sunLight(MRTFloat3 direction, MRTFloat3 color) { Light l; l.type = MRTLightTypeSunlight; l.direction = direction; l.color = color; return l; }
    static Light pointLight(MRTFloat3 position, MRTFloat3 color) { Light l; l.type = MRTLightTypePointlight; l.position = position; l.color = color; return l; }
    static Light spotLight(MRTFloat3 position, MRTFloat3 direction, float coneAngle, MRTFloat3 color) {
        Light l; l.type = MRTLightTypeSpotlight; l.position = position; l.direction = direction; l.coneAngle = coneAngle; l.color = color; return l;
    }
};

struct Submesh {                                                     // SubMesh.swift:10-33
    std::string name;
    std::vector<uint32_t> indices;                                   // 3 per triangle
    Material material{};
    size_t triangleCount() const { return indices.size() / 3; }
};

struct Mesh {                                                        // Mesh.swift:10-49
    std::vector<float> positions, normals;                           // packed xyz
    float transform[16];                                             // column-major T*R*S (Mesh.swift:24)
    std::vector<Submesh> submeshes;
};

// Directory searched for Resources/<name>.obj (Bundle.main in the reference, Model.swift:14)
inline std::string &resourceDirectory() { static std::string d = "assets/Resources"; return d; }

struct Model {                                                       // Model.swift:10-40
    std::string name;
    std::vector<Mesh> meshes;
    bool proxy = false;                                              // dragon / bunny stand-ins (absent upstream)
    Model(const std::string &name_, const float position[3], const float rotation[3], float scale) : name(name_) {
        MRTMeshData md = nullptr;
        std::string path = resourceDirectory() + "/" + name + ".obj";
        int rc = mrt_obj_load(path.c_str(), &md);
        if (rc == MRT_ERR_IO && name == "dragon") { check(mrt_dragon_proxy(&md)); proxy = true; }
        else if (rc == MRT_ERR_IO && name == "bunny") { check(mrt_bunny_proxy(&md)); proxy = true; }
        else check(rc);
        Mesh m;
        size_t nv = 0; int32_t ns = 0;
        check(mrt_meshdata_counts(md, &nv, &ns));
        m.positions.resize(nv * 3); m.normals.resize(nv * 3);
        check(mrt_meshdata_vertices(md, m.positions.data(), m.normals.data()));
        for (int32_t s = 0; s < ns; s++) {
            Submesh sm; size_t nt = 0; char buf[256];
            check(mrt_meshdata_submesh(md, s, &nt, nullptr, nullptr, nullptr, 0));
            sm.indices.resize(nt * 3);
            check(mrt_meshdata_submesh(md, s, &nt, sm.indices.data(), &sm.material, buf, sizeof buf));
            sm.name = buf;
            m.submeshes.push_back(std::move(sm));
        }
        mrt_meshdata_free(md);
        check(mrt_make_transform(position, rotation, scale, m.transform));
        meshes.push_back(std::move(m));
    }
    Model(const std::string &name_, std::initializer_list<float> position, float scale)
        : Model(name_, std::vector<float>(position).data(), std::vector<float>{0, 0, 0}.data(), scale) {}
    Model(const std::string &name_, std::initializer_list<float> position, std::initializer_list<float> rotation, float scale)
        : Model(name_, std::vector<float>(position).data(), std::vector<float>(rotation).data(), scale) {}
};

struct Scene {                                                       // Scene.swift:10-67
    std::vector<Model> models;
    Camera camera;
    std::vector<Light> lights;
    Scene(int width, int height) {
        camera = setupCamera(width, height);
        Light light1 = setupLight();
        Light light3 = Light::spotLight(f3(2, 1, 4), f3(-1.5f, -0.5f, -1.5f), 25.0f / 180.0f * 3.14159274f, f3(4, 4, 4));
        lights = {light1, light3};                                   // Scene.swift:30
    }
    virtual ~Scene() = default;
    void updateUniforms(int width, int height) { camera = setupCamera(width, height); }   // Scene.swift:36-38
    static Camera setupCamera(int width, int height) { Camera c; check(mrt_default_camera(width, height, &c)); return c; }   // :40-57
    static Light setupLight() {                                      // :59-67
        return Light::areaLight(f3(0.0f, 1.98f, 0.0f), f3(0.0f, -1.0f, 0.0f), f3(0.25f, 0.0f, 0.0f), f3(0.0f, 0.0f, 0.25f), f3(4.0f, 4.0f, 4.0f));
    }
};

struct DragonScene : Scene {                                         // DragonScene.swift:10-34
    DragonScene(int width, int height) : Scene(width, height) {
        const float pi = 3.14159274f;
        models.emplace_back("train", std::initializer_list<float>{-0.3f, 0, 0.4f}, 0.5f);
        models.emplace_back("dragon", std::initializer_list<float>{0.3f, 0.38f, 2.5f}, std::initializer_list<float>{0, pi / 2 * 1.2f, 0}, 1.2f);
        models.emplace_back("treefir", std::initializer_list<float>{0.5f, 0, -0.2f}, 0.7f);
        models.emplace_back("plane", std::initializer_list<float>{0, 0, 0}, 10.0f);
        models.emplace_back("sphere", std::initializer_list<float>{-1.9f, 0.0f, 0.3f}, 1.0f);
        models.emplace_back("sphere", std::initializer_list<float>{2.9f, 0.0f, -0.5f}, 2.0f);
        models.emplace_back("plane-back", std::initializer_list<float>{0, 0, -1.5f}, 10.0f);
    }
};

// createBuffers / geometry descriptors (Renderer.swift:107-182, Mesh.swift:39-48): hand a Scene's models and lights to an MRTScene.
// instancing: models loaded from the same resource become instances of one mesh (mrt_scene_add_instance), committed two-level.
inline void uploadScene(MRTScene dst, const Scene &scene, bool instancing = false) {
    if (instancing) check(mrt_scene_set_option(dst, "instancing", 1));
    std::vector<std::pair<std::string, int32_t>> loaded;             // resource name -> mesh id of its first use
    for (const Model &model : scene.models)
        for (const Mesh &mesh : model.meshes) {
            int32_t id = -1, source = -1;
            if (instancing && model.meshes.size() == 1) for (auto &l : loaded) if (l.first == model.name) source = l.second;
            if (source >= 0) { check(mrt_scene_add_instance(dst, source, mesh.transform, &id)); continue; }
            check(mrt_scene_add_mesh(dst, mesh.positions.data(), 12, mesh.normals.data(), 12, mesh.positions.size() / 3, mesh.transform, &id));
            for (const Submesh &sm : mesh.submeshes) check(mrt_mesh_add_submesh(dst, id, sm.indices.data(), sm.triangleCount(), &sm.material, nullptr));
            loaded.emplace_back(model.name, id);
        }
    check(mrt_scene_set_lights(dst, scene.lights.data(), (int32_t)scene.lights.size()));
}

class Renderer {                                                     // Renderer.swift:12-357
  public:
    static constexpr int maxFramesInFlight = 3;                      // Renderer.swift:33 (here: setOption("frames_in_flight", n); library default: the same three, each a pass of up to 8 frames)
    // instancing = true: models loaded from the same resource become instances of one mesh (mrt_scene_add_instance) and the scene is committed
    // two-level — one BLAS per distinct mesh + a TLAS (the reference's instance acceleration structure, Renderer.swift:193-213)
    Renderer(int width, int height, const Scene &scene, int device = 0, uint32_t seed = 1, int max_bounces = 3, bool instancing = false) : w_(width), h_(height) {
        check(mrt_context_create(device, &ctx_));                    // MTLCreateSystemDefaultDevice + queue (:46-59)
        try {
            check(mrt_scene_create(ctx_, &scene_));
            uploadScene(scene_, scene, instancing);                  // createBuffers / geometry descriptors (:107-182, Mesh.swift:39-48)
            check(mrt_scene_commit(scene_));                         // createAccelerationStructures (:184-214)
            check(mrt_renderer_create(ctx_, scene_, width, height, seed, max_bounces, &r_));
            check(mrt_renderer_set_camera(r_, &scene.camera));
        } catch (...) { destroy(); throw; }
    }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;
    ~Renderer() { destroy(); }
    void draw(int frames = 1) { check(mrt_renderer_render(r_, frames)); }                   // draw(in:) (:284-351)
    void wait() { check(mrt_renderer_wait(r_)); }
    uint64_t framesCompleted() const { uint64_t f = 0; check(mrt_renderer_frames_completed(r_, &f)); return f; }   // the completion handler of :285-287 as a poll; never blocks
    // updateUniforms (:216-229): size, frameIndex, lightCount and camera as the one 96-byte block the reference binds at buffer index 0
    MRTUniforms uniforms() const { MRTUniforms u; check(mrt_renderer_get_uniforms(r_, &u)); return u; }
    void setUniforms(const MRTUniforms &u) { check(mrt_renderer_set_uniforms(r_, &u)); w_ = u.width; h_ = u.height; }
    void drawableSizeWillChange(int width, int height) { w_ = width; h_ = height; check(mrt_renderer_resize(r_, width, height)); }   // :353-356
    uint32_t frameIndex() const { uint32_t f = 0; check(mrt_renderer_frame_index(r_, &f)); return f; }
    void setFrameIndex(uint32_t f) { check(mrt_renderer_set_frame_index(r_, f)); }
    // animated transforms: new object->world matrix (column-major 4x4) for one mesh / instance, then commit(); a two-level scene rebuilds only its TLAS
    void setInstanceTransform(int32_t meshId, const float transform[16]) { check(mrt_scene_set_instance_transform(scene_, meshId, transform)); }
    /// deforming geometry: new object-space positions / normals (packed xyz) of one mesh's vertices, same count; commit() then refits the tree of a flattened scene
    void updateMesh(int32_t meshId, const float *positions, const float *normals, size_t vertexCount) { check(mrt_scene_update_mesh(scene_, meshId, positions, 12, normals, 12, vertexCount)); }
    // the checked form: one normal per vertex, or std::invalid_argument before anything is read
    void updateMesh(int32_t meshId, const std::vector<float> &positions, const std::vector<float> &normals) {
        if (positions.size() % 3 != 0 || normals.size() != positions.size()) throw std::invalid_argument("updateMesh: positions and normals must hold 3 floats per vertex, the same number of vertices");
        updateMesh(meshId, positions.data(), normals.data(), positions.size() / 3);
    }
    void commit() { check(mrt_scene_commit(scene_)); }
    // implementation knobs (mrt_abi.h): "frames_in_flight" (HIP streams, default 12), "frame_batch" (frames per pass, default 4), ...
    void setOption(const char *key, double value) { check(mrt_renderer_set_option(r_, key, value)); }
    double option(const char *key) const { double v = 0; check(mrt_renderer_get_option(r_, key, &v)); return v; }
    std::vector<float> accumulation() { std::vector<float> a((size_t)w_ * h_ * 4); check(mrt_renderer_read_accum(r_, a.data(), a.size() * 4)); return a; }
    std::vector<uint8_t> tonemapped() { std::vector<uint8_t> a((size_t)w_ * h_ * 4); check(mrt_renderer_read_tonemapped_rgba8(r_, a.data(), a.size())); return a; }
    MRTRenderStats stats() { MRTRenderStats s; check(mrt_renderer_stats(r_, &s)); return s; }
    MRTSceneStats sceneStats() { MRTSceneStats s; check(mrt_scene_stats(scene_, &s)); return s; }
    int width() const { return w_; }
    int height() const { return h_; }

  private:
    void destroy() {
        if (r_) mrt_renderer_destroy(r_);
        if (scene_) mrt_scene_destroy(scene_);
        if (ctx_) mrt_context_destroy(ctx_);
        r_ = nullptr; scene_ = nullptr; ctx_ = nullptr;
    }
    int w_, h_;
    MRTContext ctx_ = nullptr;
    MRTScene scene_ = nullptr;
    MRTRenderer r_ = nullptr;
};

// The same Renderer over the n GPUs of one node (one process): the scene is replicated, the image sharded by 8x8 screen tile
// (tile_id % n == rank), and gather() runs the ONE reduce(sum) per output image (RCCL over xGMI).  The reference creates a single
// MTLDevice (Renderer.swift:46-59); this widens that seam.  The assembled image is bit-identical to Renderer's.
class GroupRenderer {
  public:
    GroupRenderer(int width, int height, const Scene &scene, const std::vector<int> &devices, uint32_t seed = 1, int max_bounces = 3, bool instancing = false) : w_(width), h_(height) {
        check(mrt_group_create(devices.data(), (int32_t)devices.size(), &g_));
        try {
            MRTContext c0 = nullptr; check(mrt_group_context(g_, 0, &c0));
            check(mrt_scene_create(c0, &template_));
            uploadScene(template_, scene, instancing);               // the template is replicated and committed on every device
            check(mrt_group_renderer_create(g_, template_, width, height, seed, max_bounces, &gr_));
            check(mrt_group_set_camera(gr_, &scene.camera));
        } catch (...) { destroy(); throw; }
    }
    GroupRenderer(const GroupRenderer &) = delete;
    GroupRenderer &operator=(const GroupRenderer &) = delete;
    ~GroupRenderer() { destroy(); }
    int size() const { int32_t n = 0; check(mrt_group_size(g_, &n)); return n; }
    void draw(int frames = 1) { check(mrt_group_render(gr_, frames)); }                      // every device; returns at once
    void wait() { check(mrt_group_wait(gr_)); }
    uint64_t framesCompleted() const { uint64_t f = 0; check(mrt_group_frames_completed(gr_, &f)); return f; }
    void setOption(const char *key, double value) { check(mrt_group_set_option(gr_, key, value)); }
    std::vector<float> gather() { std::vector<float> a((size_t)w_ * h_ * 4); check(mrt_group_gather(gr_, a.data(), a.size() * 4)); return a; }   // the assembled image
    MRTRenderStats stats() { MRTRenderStats s; check(mrt_group_stats(gr_, &s)); return s; }
    std::string reduceMode() const { int32_t m = 0; char note[256]; check(mrt_group_reduce_mode(g_, &m, note, sizeof note)); return note; }

  private:
    void destroy() {
        if (gr_) mrt_group_renderer_destroy(gr_);
        if (template_) mrt_scene_destroy(template_);
        if (g_) mrt_group_destroy(g_);
        gr_ = nullptr; template_ = nullptr; g_ = nullptr;
    }
    int w_, h_;
    MRTGroup g_ = nullptr;
    MRTScene template_ = nullptr;
    MRTGroupRenderer gr_ = nullptr;
};

}  // namespace mrt
