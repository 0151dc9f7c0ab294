// scene_device.h — HBM data layout of a committed scene, shared by the BVH builder (bvh_build.hip)
// and the render kernels (renderer.hip, traverse.h, traverse_wide.h).  DESIGN.md §4 documents every array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/mrt_abi.h"

namespace mrt {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define MRT_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) return ::mrt::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

// ---- traversal node: 64 B, one per box ("rope" layout: every node carries, for each of the 8
// ray-direction octants, the node to continue with once its subtree is finished).
//   word 0..2  lo.xyz     word 3  a : leaf ? (NODE_LEAF | first packet) : index of the left child
//   word 4..6  hi.xyz     word 7  b : leaf ? triangle count : (near-mask << 24) | index of the right child
//   word 8..15 esc[8]                 : next node per octant, NODE_TERM = traversal finished
// near-mask bit o = 1 when the RIGHT child is visited first by rays of octant o
// (octant bit k set <=> direction component k negative).
constexpr uint32_t NODE_LEAF = 0x80000000u;
constexpr uint32_t NODE_TERM = 0xFFFFFFFFu;
constexpr uint32_t NODE_INDEX_MASK = 0x00FFFFFFu;

// ---- triangle packet: 48 B, leaf-contiguous, already in world space and in Möller–Trumbore form
//   {v0.xyz, gid} {e1.xyz, -} {e2.xyz, -}
// ---- shading record per global triangle id (16 B): vertex ids into `normals` + (instance<<16 | geometry)

// ---- wide node: 80 B = 5 x float4, eight children with 8-bit quantised boxes (after Ylitie, Karras, Laine
// 2017, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs"):
//   f4 0: origin p.xyz | ex | ey<<8 | ez<<16 | imask<<24      child box = p + q * 2^e (e: int8); imask bit i = child i is a node
//   f4 1: child_base | tri_base | meta[0..3] | meta[4..7]      node child i -> child_base + popc(imask & ((1<<i)-1));
//                                                              leaf child i -> packets tri_base + (meta&31) .. + (meta>>5) - 1
//   f4 2: qlo_x[8] qlo_y[8]      f4 3: qlo_z[8] qhi_x[8]      f4 4: qhi_y[8] qhi_z[8]      (empty slot: qlo=255, qhi=0)
// Slot bit k set = the child lies on the + side of the node centre along axis k, so children are entered
// front to back in the order of (slot ^ ray octant).
// (A six-children-in-64-bytes variant of this node, -DMRT_WIDE6, measured worse in round 4 and was removed in round 6: profiles/r04_wide6_ab.txt.)
#ifndef MRT_WNODE_STRIDE
#define MRT_WNODE_STRIDE 5
#endif
constexpr int WIDE_N = 8;                             // children per wide node
constexpr uint32_t WNODE_STRIDE = MRT_WNODE_STRIDE;   // float4 units between wide nodes in HBM (5 = packed 80 B; 8 = one 128-B line each)
#ifndef MRT_WPACKET_STRIDE
#define MRT_WPACKET_STRIDE 3
#endif
constexpr uint32_t WPK = MRT_WPACKET_STRIDE;          // float4 units between the triangle packets of the 8-wide layout (3 = packed 48 B; 4 = one 64-byte sector each, never straddling two)
constexpr int WIDE_STACK_MAX = 96;   // deepest 8-wide tree the traversal kernels walk: their LDS stack is sized per launch from the scene's depth (320 B per wave and level: 4.2 KB at DragonScene's 13
                                     // levels, 7.7 KB at 24 — five instead of six waves per SIMD —, 30 KB at 96: what a radix tree over 63-bit keys + 26 levels of index splits can need at most); the agglomerative
                                     // builder's deeper trees (a chain of nested triangles) are built again as radix trees (build_flat); only a tree deeper than this keeps the rope layout, and MRTSceneStats::wide_layout says so
constexpr int WIDE_DEPTH_REBUILD = 48;     // an 8-wide tree from the agglomerative builder deeper than this is built again from the radix tree (build_flat)
constexpr int WIDE_STACK_TWO_LEVEL = 30;   // two-level scenes: TLAS levels + 1 (the TLAS group parked at instance entry) + the deepest BLAS; 5-bit depth fields
constexpr uint32_t WIDE_WORLD_RAY_BYTES = 6 * 64 * 4;   // two-level stream traversal: the world-space ray of every lane (o.xyz, d.xyz) parked in LDS in front of the stack
constexpr uint32_t WIDE_STACK_LEVEL_BYTES = 320;   // per wave and level: 64 x 4 B {child_base << 8 | hit bits} + 64 x 1 B {imask}

struct LightDev {            // 96 B, derived once per mrt_scene_set_lights from the 128-B MRTLight
    float4 position;         // .w = type (as int bits)
    float4 color;
    float4 forward;          // area light (un-normalised, as the reference uses it)
    float4 right;
    float4 up;
    float4 dirn;             // spot: normalize(direction), .w = cos(coneAngle); sun: normalize(direction)
};

// ---- two-level scenes (scene option instancing = 1; two_level.hip): one BLAS per distinct mesh, built in OBJECT space by the same
// pipeline as the flattened scene, one record per instance, and a TLAS (rope nodes again, built on the host over the instances'
// world boxes) whose leaves are ranges of `tlas_index`.  Replaces MTLAccelerationStructureInstanceDescriptor + the instance
// acceleration structure of Renderer.swift:193-213.  A ray is taken into object space at the instance boundary WITHOUT
// renormalising its direction, so the hit distance t is the world-space t.
struct InstanceDev {          // 80 B
    float4 w2o[3];            // rows of the world->object 3x4: p_obj.k = fma(r.z, p.z, fma(r.y, p.y, r.x * p.x)) + r.w (directions: without r.w)
    uint32_t node_base;       // first rope node of its BLAS in `bnodes`
    uint32_t packet_base;     // first triangle packet of its BLAS in `bpackets`
    uint32_t gid_base;        // global id of its first triangle: instance-major, then geometry, then primitive — the numbering of the flattened scene
    uint32_t ts_base;         // first shading record of its BLAS in `tri_shade`
    uint32_t vbase;           // first vertex of its mesh in `normals`
    uint32_t ntri, blas;
    uint32_t wroot;           // root of its BLAS in the shared 8-wide array `wnodes` (child / packet indices in there are absolute)
};

struct SceneView {           // passed by value to kernels
    const float4 *nodes;         // 4 x float4 per node
    const float4 *packets;       // 3 x float4 per triangle, leaf order
    const uint4 *tri_shade;      // per gid
    const float4 *normals;       // object space, concatenated over meshes (float3 stride 16, Mesh.swift:27-29)
    const float4 *base_color;    // per resource slot = instance*max_sub + geometry (Renderer.swift:139)
    const float4 *materials;     // per resource slot, 3 x float4: baseColor | dissolve, specular | specularExponent, emission | refractionIndex (materials extension)
    const float4 *inst_cols;     // 3 x float4 per instance: columns 0..2 of the 4x3 transform
    const uint32_t *geom_base;   // per resource slot: first gid
    const LightDev *lights;
    // 8-wide compressed layout (WideNode below) for the LDS-stack traversal backend; num_wnodes == 0 → not built
    const float4 *wnodes;        // 5 x float4 per wide node
    const float4 *wpackets;      // 3 x float4 per triangle, grouped per wide node
    // two-level scenes: `nodes` is the TLAS; num_inst == 0 -> flattened scene
    const InstanceDev *inst;
    const uint32_t *tlas_index;  // instance ids, TLAS leaf order
    const uint32_t *wtlas_index; // instance ids in the leaf order of the 8-wide TLAS (wnodes[0 ..]: its leaf children are single instances)
    const float4 *inst_box;      // two-level scenes: per instance four float4 — the box of its BLAS in OBJECT space (lo, hi: what the BLAS pass tests a fetched pair's ray against) and the instance's padded WORLD box (lo, hi: what the flat TLAS pass, k_tl_top_flat, queues pairs on)
    const uint32_t *tri_packet;  // two-level scenes with the 8-wide layout: per shading record (InstanceDev::ts_base + triangle of the BLAS) its packet in wpackets — how k_shade<.., PAIRS> finds the triangle a key names
    const float4 *bnodes;        // rope nodes of all BLASes, 4 x float4 each; followed, in the same allocation, by
    const float4 *bpackets;      // their triangle packets (object space), 3 x float4 each
    uint32_t num_inst;
    uint32_t num_wpackets, num_wtlas;   // entries of wpackets / wtlas_index (the -DMRT_DEBUG_BOUNDS build checks every index against these)
    uint32_t num_wnodes;
    uint32_t num_nodes;
    uint32_t num_tris;
    int32_t light_count;
    int32_t max_sub;
};

// Scratch memory of one build: a few large allocations handed out in 256-byte-aligned pieces and freed together (a build used to make ~60 hipMalloc / hipFree
// pairs, each a few tens of microseconds and the frees device-synchronising).  Declare it before the buffers that borrow from it.
struct ScratchArena {
    std::vector<void *> chunks; uint8_t *cur = nullptr; size_t left = 0, chunk_bytes = (size_t)1 << 20;
    ScratchArena() = default;
    ScratchArena(const ScratchArena &) = delete; ScratchArena &operator=(const ScratchArena &) = delete;
    ~ScratchArena() { for (void *c : chunks) (void)hipFree(c); }
    hipError_t take(void **out, size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > left) {
            const size_t c = bytes > chunk_bytes ? bytes : chunk_bytes;
            void *m = nullptr;
            hipError_t e = hipMalloc(&m, c);
            if (e != hipSuccess) return e;
            chunks.push_back(m); cur = (uint8_t *)m; left = c;
        }
        *out = cur; cur += bytes; left -= bytes;
#ifdef MRT_POISON_ALLOC      // diagnostics build: no allocation starts out as zeros (every dword = 1: a counter that was assumed 0 is off by one, an index stays inside its array)
        (void)hipDeviceSynchronize(); (void)hipMemsetD32((hipDeviceptr_t)*out, 1, bytes / 4); (void)hipDeviceSynchronize();
#endif
        return hipSuccess;
    }
};

template <class T> struct DevBuf {
    T *p = nullptr; size_t n = 0; bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; n = 0; owned = true; }
    hipError_t alloc(size_t count, unsigned flags = 0 /* hipDeviceMallocDefault; hipDeviceMallocUncached for streamed buffers (experiment) */) {
        if (count == 0) count = 1;
        if (p && owned && n == count && !flags) return hipSuccess;          // same size as before (a re-commit, a resize back): keep the allocation, the caller overwrites it
        release();
        hipError_t e = flags ? hipExtMallocWithFlags((void **)&p, count * sizeof(T), flags) : hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count; else p = nullptr;
#ifdef MRT_POISON_ALLOC
        if (e == hipSuccess) { (void)hipDeviceSynchronize(); (void)hipMemsetD32((hipDeviceptr_t)p, 1, count * sizeof(T) / 4); (void)hipDeviceSynchronize(); }
#endif
        return e;
    }
    hipError_t alloc_in(ScratchArena &a, size_t count) {       // a piece of the arena: not freed here, gone with the arena
        release();
        if (count == 0) count = 1;
        hipError_t e = a.take((void **)&p, count * sizeof(T));
        if (e == hipSuccess) { n = count; owned = false; } else p = nullptr;
        return e;
    }
    size_t bytes() const { return n * sizeof(T); }
};

// Pinned host memory that only grows: the staging area of a scene's uploads.  A commit concatenates the caller's arrays straight into it (one pass, no zero fill)
// and the copies to the device are DMA from pinned pages; it stays with the scene, so the commits of an animated scene pay no allocation.
struct PinnedBuf {
    void *p = nullptr; size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete; PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = (bytes + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
};

struct HostMesh {                  // what the caller handed over through mrt_scene_add_mesh / _add_submesh / _add_instance
    int source = -1;               // >= 0: an instance of that mesh (shares its vertex arrays, submeshes and materials; its own arrays stay empty)
    std::vector<float> positions;  // packed xyz, object space
    std::vector<float> normals;    // packed xyz, object space
    float xf[16];                  // column-major, last row forced to (0,0,0,1)
    std::vector<std::vector<uint32_t>> sub_indices;
    std::vector<MRTMaterial> sub_materials;
    bool dirty = false;            // vertices replaced by mrt_scene_update_mesh since the last commit (a refit recomputes the boxes of such meshes' triangles only)
};

struct BuildOptions {
    int builder = 1;          // 0 = Karras radix tree (plain LBVH), 1 = PLOC agglomeration over the Morton order, 2 = binned SAH built on the host (quality yardstick, seconds)
    int max_leaf = 4;         // SAH leaf collapse limit
    float cost_trav = 1.0f;   // SAH constants
    float cost_isect = 1.0f;
    int ploc_radius = 16;
    int wide = 1;             // build the 8-wide compressed layout: every ray of the pipeline walks it
    int rope = 0;             // 1: also emit the rope layout (binary nodes with escape links + its own copy of the packets); 0: only when the scene cannot have the 8-wide layout
    int wide_collapse = 1;    // 8-wide layout: 0 = greedy collapse of the binary tree (largest child first), 1 = SAH-optimal collapse by dynamic programming (k_wide_dp)
    float wide_cost_node = 1.0f, wide_cost_tri = 0.5f;      // its constants: a node visit (eight box tests + an iteration) against one triangle test — ~250 against ~125 VALU instructions in the stream loop; 0.3 until round 5
                                                            // (0.45 ... 1.0 measured +1 ... +2 % on DragonScene, 0.5 against 0.3: garden 4K +1 %, dragon x 4 +3 ... +4 %, hostile +1.5 %, Cornell 256^2 +8 ... +13 %; profiles/r05_wide_cost_tri.txt)
    float presplit = 4.0f;    // > 0: a triangle whose box is longer than presplit x the mean triangle extent enters the build as several references (k_split_emit); 0 = off
    int refit = 1;            // a commit after mrt_scene_update_mesh alone (same topology, new vertex positions / normals) REFITS the 8-wide tree of a flattened scene — packets rewritten, boxes recomputed bottom-up, the tree's shape kept — instead of building it again; 0: always build
    float refit_max_cost_ratio = 0.0f;      // > 0: a commit whose refit leaves MRTSceneStats.wide_cost above this factor of wide_cost_built builds the tree again instead (0 = the caller decides, from those two numbers)
    int refit_fenced = 0;     // 1: the bottom-up pass of the build with __threadfence() hand-offs instead of write-through stores (the slow reference form; same tree bit for bit)
    int validate = 1;         // check every index of the committed layout on the host (validate_layout), once per commit
    int instancing = 0;       // 0: flatten every instance into one world-space BVH (default; the reference never shares a primitive AS);
                              // 1: two-level — a BLAS per distinct mesh shared by its instances + a TLAS; transform changes rebuild only the TLAS
};

// One BLAS of a two-level scene inside the scene's shared arrays: what a refit of that BLAS (mrt_scene_update_mesh on its mesh + commit) needs to find it again
struct BlasRange {
    uint32_t src_mesh = 0;                                   // the mesh it was built from (index into the scene's mesh list)
    uint32_t wnode_base = 0, wnodes = 0;                     // its 8-wide nodes in `wnodes` (absolute child / packet indices inside)
    uint32_t packet_base = 0, ntri = 0;                      // its packets in `wpackets` and in `bpackets` (one per triangle: BLASes are built without pre-splitting)
    uint32_t node_base = 0, rope_nodes = 0;                  // its rope nodes in `bnodes`
    uint32_t ts_base = 0, vbase = 0;
    std::vector<uint32_t> wide_levels;                       // nodes per level of its 8-wide tree (BFS numbering)
    float wide_cost_built = 0.0f, sah_cost_built = 0.0f;     // as its build left them (MRTSceneStats)
    float wide_cost = 0.0f, leaf_growth = 1.0f;              // leaf_growth: its leaf boxes' area against the build's, chained over its refits
};

struct DeviceScene {
    DevBuf<float4> nodes, packets, normals, base_color, materials, inst_cols, wnodes, wpackets;
    uint32_t num_wnodes = 0; int wide_depth = 0;
    std::vector<uint32_t> wide_levels;           // flattened scenes: nodes per level of the 8-wide tree (BFS numbering) — what a refit walks bottom-up (build_flat, refit)
    uint32_t refits = 0;                         // commits served by a refit since the last build
    float sah_cost_built = 0.0f;                 // stats.sah_cost as the last build left it (a refit scales it by the 8-wide tree's cost ratio)
    uint64_t refit_triangles = 0;                // triangles of the build that made the 8-wide layout (a refit needs the same count)
    uint32_t num_packets = 0;        // triangle packets per layout = build references (stats.triangles, or more when long triangles were pre-split)
    uint32_t rope_nodes = 0;         // surviving rope nodes (stats.bvh_nodes reports the 8-wide node count when that layout is built)
    size_t packets_offset = 0;       // packets start at nodes.p + packets_offset (float4 units); `packets` itself is unused
    DevBuf<uint4> tri_shade;
    DevBuf<uint32_t> geom_base;
    DevBuf<LightDev> lights;
    int light_count = 0;
    MRTSceneStats stats{};
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};     // box of the whole tree (padded leaf boxes)
    // two-level scenes (two_level.hip)
    DevBuf<InstanceDev> inst; DevBuf<uint32_t> tlas_index, wtlas_index, tri_packet; DevBuf<float4> bnodes, inst_box;
    uint32_t tlas_wcap = 0; int blas_wdepth = 0;               // 8-wide layout: node slots reserved for the TLAS in front of the BLASes, deepest BLAS (0: rope only)
    size_t bpackets_offset = 0; uint32_t num_inst = 0;
    std::vector<InstanceDev> h_inst;                           // host copy: transform updates rewrite the rows and rebuild the TLAS only
    std::vector<BlasRange> blas_ranges;                        // per BLAS: where it lies in the shared arrays (refit_two_level)
    bool blas_all_wide = false;                                // every BLAS has the 8-wide layout (the shared wnodes / wpackets exist)
    std::vector<float> blas_lo, blas_hi;                       // per BLAS root box (object space), 3 floats each
    float tlas_ms = 0;                                         // host + upload time of the last TLAS build
    bool validate = true, validated_blas = false;              // commit-time index validation (two_level.hip validate_layout); BLAS part already checked
    DevBuf<float> g_pos; DevBuf<uint32_t> g_idx, g_recs;       // flattened build: packed object-space positions, indices and submesh records as uploaded (kept: a commit that only changes transforms does not upload them again)
    PinnedBuf stage;                                           // upload staging of build_flat (grow-only, reused by every commit of this scene)
    double commit_ms[6] = {0, 0, 0, 0, 0, 0};                    // host wall time of the last flat build, by phase: staging (reserve + fill), device allocations + upload enqueue, topology (sort .. refit, incl. its read-backs), 8-wide emit, rope emit, validation (mrt_debug_commit_times)
    SceneView view() const;
};

// bvh_build.hip
void pack_material(const MRTMaterial &m, float4 *out3);
int wide_histogram(const DeviceScene &sc, hipStream_t stream, uint32_t out12[12]);      // diagnostics: children per 8-wide node
int wide_tree_cost(const float4 *wnodes, uint32_t first, uint32_t count, uint32_t root, float c_node, float c_tri, hipStream_t stream, float *out);      // SAH cost of the 8-wide subtree [first, first + count) rooted at `root`, as it lies in memory, per unit of root area
int layout_limits(uint64_t triangles, uint64_t nodes);    // MRT_OK, or MRT_ERR_UNSUPPORTED when the traversal layouts cannot address such a scene
int build_scene(const std::vector<HostMesh> &meshes, const BuildOptions &opt, hipStream_t stream, DeviceScene &out, bool only_transforms_changed = false, bool only_vertices_changed = false);      // only_transforms_changed: same meshes, submeshes and options as the commit before (flattened scenes keep their geometry on the device)
// bvh_host_sah.cpp (builder = 2): binned-SAH topology over n reference boxes, built on the host
void host_sah_topology(const float4 *lo, const float4 *hi, uint32_t n, std::vector<uint32_t> &order, std::vector<uint32_t> &left, std::vector<uint32_t> &right, std::vector<uint32_t> &parent);
struct MeshRef { const HostMesh *g; const float *xf; };         // geometry + object->world matrix (column-major 4x4)
int build_flat(const std::vector<MeshRef> &refs, const BuildOptions &opt, hipStream_t stream, DeviceScene &out, PinnedBuf *stage = nullptr, bool geometry_unchanged = false, bool refit = false);      // stage: the upload staging to use instead of out.stage (the BLAS builds of a two-level scene share their scene's)
// two_level.hip
int build_two_level(const std::vector<HostMesh> &meshes, const BuildOptions &opt, hipStream_t stream, DeviceScene &out);
int refit_two_level(const std::vector<HostMesh> &meshes, const BuildOptions &opt, hipStream_t stream, DeviceScene &out);      // after mrt_scene_update_mesh alone: the BLASes of the updated meshes refitted in place (both layouts) + the TLAS; MRT_ERR_UNSUPPORTED (no message): the scene cannot be refitted, build it
int refit_blas(const HostMesh &g, const BlasRange &br, hipStream_t stream, DeviceScene &out, float root_lo[3], float root_hi[3], float *ms_out, float *growth_out);      // bvh_build.hip
int update_tlas(const std::vector<HostMesh> &meshes, hipStream_t stream, DeviceScene &out);      // after transform changes: instance rows + TLAS, BLASes untouched
int validate_layout(const DeviceScene &sc, hipStream_t stream, bool tlas_only, const float4 *wnodes_override = nullptr);      // wnodes_override: a device copy of the 8-wide nodes to check in place of the scene's (mrt_debug_validate_patched)
         // every index of the 8-wide layout / instance rows inside its array; MRT_ERR_STATE + message otherwise
int upload_lights(const MRTLight *lights, int count, hipStream_t stream, DeviceScene &out);

}  // namespace mrt
