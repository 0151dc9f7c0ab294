// api_types.h — what the opaque handles of include/mrt_abi.h point at (shared by api.cpp and group.hip).
#pragma once
#include "renderer.h"

struct MRTContext_ {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char name[256] = {0};
    int live = 0;                           // scenes and renderers made on this context and not yet destroyed: mrt_context_destroy refuses while any is left
};
struct MRTScene_ {
    MRTContext ctx = nullptr;
    std::vector<mrt::HostMesh> meshes;
    std::vector<MRTLight> lights;
    mrt::BuildOptions opt;
    mrt::DeviceScene dev;
    bool committed = false;
    bool only_vertices_changed = false;     // since the last commit: nothing but mrt_scene_update_mesh calls — a flattened scene with the 8-wide layout refits its tree (scene option refit)
    bool only_transforms_changed = false;   // since the last commit: a two-level scene rebuilds its TLAS only, a flattened one rebuilds from the geometry already on the device
    int renderers = 0;                      // renderers made on this scene and not yet destroyed: mrt_scene_destroy refuses while any is left (they hold a pointer to `dev`)
    size_t stage_need = 0;                  // bytes the upload staging of the meshes added so far will take (mrt_mesh_add_submesh grows the pinned area)
};
struct MRTRenderer_ {
    MRTContext ctx = nullptr;
    MRTScene scene = nullptr;
    mrt::Renderer r;
};
struct MRTMeshData_ { mrt::MeshData m; };


#define MRT_TRY try {
#define MRT_CATCH                                                                      \
    } catch (const std::bad_alloc &) { mrt::set_error("out of host memory"); return MRT_ERR_OUT_OF_MEMORY; } \
    catch (const std::exception &e) { mrt::set_error(std::string("exception: ") + e.what()); return MRT_ERR_INVALID_ARGUMENT; } \
    catch (...) { mrt::set_error("unknown exception"); return MRT_ERR_INVALID_ARGUMENT; }
#define REQUIRE(cond, msg) do { if (!(cond)) { mrt::set_error(msg); return MRT_ERR_INVALID_ARGUMENT; } } while (0)
