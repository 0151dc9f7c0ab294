/*
 * mrt_abi.h — C ABI of the MI355X-native path tracer (libmrt_hip.so).
 *
 * This is the drop-in boundary for the hot path of JaapWijnen/metal-raytracing: everything
 * `Renderer.draw(in:)` binds to `raytracingKernel` (Renderer.swift:302-329), the data contract
 * of the bridging header (ShaderTypes.h:60-107) and the acceleration-structure build
 * (Renderer.swift:184-214, Utilities.swift:29-85).  The reference has no FFI of its own; each
 * entry point below cites the reference interface it replaces.  Plain pointers and sizes only:
 * no C++ types, no torch types.  Every call returns an int status (MRT_OK == 0) and never
 * throws or aborts across the boundary; mrt_last_error() gives the thread-local message.
 *
 * Threading: one thread drives a renderer at a time (the reference drives Renderer from the
 * main thread only, Renderer.swift:284).  The library never calls back into the caller.
 */
#ifndef MRT_ABI_H
#define MRT_ABI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRT_ABI_VERSION 3      /* 3: MRTSceneStats grew (wide_cost, wide_cost_built, refits: what a refit did to the tree); 2: (wide_layout, wide_depth), the renderer's option keys split into mrt_renderer_set_option (six host keys) and mrt_debug_renderer_set_option */

/* ---------------------------------------------------------------- status codes */
enum {
    MRT_OK = 0,
    MRT_ERR_INVALID_ARGUMENT = 1,
    MRT_ERR_NO_DEVICE = 2,      /* no HIP device / HIP runtime failure at context creation   */
    MRT_ERR_HIP = 3,            /* a HIP call failed; message carries hipGetErrorString       */
    MRT_ERR_IO = 4,             /* OBJ/MTL file could not be read                             */
    MRT_ERR_STATE = 5,          /* call order violated (e.g. render before scene commit)      */
    MRT_ERR_OUT_OF_MEMORY = 6,
    MRT_ERR_UNSUPPORTED = 7
};

/* ---------------------------------------------------------------- data contract
 * Bit layout of ShaderTypes.h:60-107.  `vector_float3` is 16-byte sized and aligned, hence the
 * explicit pad word.  Sizes and offsets are static_assert-ed in metal-raytracing_amd/csrc/abi_check.h
 * (compiled into the library) and checked through ctypes in tests/test_abi_and_host.py.         */
typedef struct { float x, y, z, _pad; } MRTFloat3;                    /* simd vector_float3     */

typedef struct {                                                      /* ShaderTypes.h:60-65    */
    MRTFloat3 position, right, up, forward;
} MRTCamera;                                                          /* 64 B                   */

typedef enum {                                                        /* ShaderTypes.h:67-74    */
    MRTLightTypeUnused = 0, MRTLightTypeSunlight = 1, MRTLightTypeSpotlight = 2,
    MRTLightTypePointlight = 3, MRTLightTypeAreaLight = 4
} MRTLightType;

typedef struct {                                                      /* ShaderTypes.h:76-87    */
    int32_t   type;            /* @0   (NSInteger on the Swift side: low 32 bits, LE)          */
    int32_t   _pad0[3];
    MRTFloat3 position;        /* @16                                                          */
    MRTFloat3 color;           /* @32                                                          */
    MRTFloat3 forward;         /* @48  area light                                              */
    MRTFloat3 right;           /* @64                                                          */
    MRTFloat3 up;              /* @80                                                          */
    float     coneAngle;       /* @96  spot light                                              */
    float     _pad1[3];
    MRTFloat3 direction;       /* @112                                                         */
} MRTLight;                                                           /* 128 B                  */

typedef struct {                                                      /* ShaderTypes.h:89-97    */
    int32_t   width, height, blocksWide;
    uint32_t  frameIndex;
    int32_t   lightCount;
    int32_t   _pad[3];
    MRTCamera camera;          /* @32                                                          */
} MRTUniforms;                                                        /* 96 B                   */

typedef struct {                                                      /* ShaderTypes.h:99-107   */
    MRTFloat3 baseColor;       /* @0  — the only field raytracingKernel reads (:269)           */
    MRTFloat3 specular;        /* @16                                                          */
    MRTFloat3 emission;        /* @32                                                          */
    float     specularExponent;/* @48                                                          */
    float     refractionIndex; /* @52                                                          */
    float     dissolve;        /* @56                                                          */
    float     _pad;
} MRTMaterial;                                                        /* 64 B                   */

/* One ray / one intersector result, as `metal::raytracing::ray` and
 * `intersector<triangle_data, instancing>::result_type` present them (Raytracing.metal:214-221,
 * :230-247).  Used by the query entry points (parity tests, brute-force cross-checks).         */
typedef struct {
    float origin[3];    float min_distance;
    float direction[3]; float max_distance;
} MRTRay;                                                             /* 32 B                   */

typedef struct {
    int32_t type;              /* 0 = none, 1 = triangle                                       */
    float   distance;
    int32_t instance_id;       /* mesh index in scene order (Renderer.swift:193-195)           */
    int32_t geometry_id;       /* submesh index within the mesh (Mesh.swift:39-48)             */
    int32_t primitive_id;      /* triangle index within the submesh                            */
    float   u, v;              /* triangle_barycentric_coord: weights of vertices 1 and 2      */
    int32_t _pad;
} MRTIntersection;                                                    /* 32 B                   */

typedef struct {
    uint64_t triangles;        /* T: triangles in the committed scene                          */
    uint64_t vertices;         /* V                                                            */
    uint64_t bvh_nodes;        /* nodes in the traversal layout                                */
    uint64_t bvh_leaves;
    uint64_t scene_bytes;      /* device bytes of nodes + triangle packets + shading tables    */
    float    build_ms;         /* device time of the last mrt_scene_commit                     */
    float    sah_cost;         /* SAH cost of the emitted tree (Ct=1, Ci=1); after a refit: the build's figure x wide_cost / wide_cost_built */
    int32_t  instances;
    int32_t  max_submeshes;    /* resource-table stride (Renderer.swift:128-139)               */
    int32_t  max_leaf_tris;
    int32_t  max_depth;
    int32_t  wide_layout;      /* 1: the scene has the 8-wide layout and every ray walks it; 0: it could not be built (tree deeper than the
                                  traversal stack can be made, node index beyond 24 bits, scene option wide = 0) and every ray falls back to
                                  the binary rope walk — about a third of the rate                                                        */
    int32_t  wide_depth;       /* levels of the 8-wide tree (two-level scenes: TLAS levels + 1 + the deepest BLAS); the traversal kernels'
                                  LDS stack is sized from it at every launch: 320 B per wave and level                                     */
    float    wide_cost;        /* SAH cost of the 8-wide tree AS IT LIES IN MEMORY, per unit of root area: sum over child boxes (decoded as the
                                  traversal decodes them) of area x (node cost | triangle cost x triangles).  Recomputed by every build and every
                                  refit; two-level scenes: mean over the BLASes.  0 without the 8-wide layout                              */
    float    wide_cost_built;  /* the same as the last BUILD left it: wide_cost / wide_cost_built is how much refits have loosened the tree —
                                  the signal to build again (scene option "refit_max_cost_ratio" does it by itself)                         */
    uint32_t refits;           /* commits served by a refit since the last build                                                          */
    float    leaf_growth;      /* surface area of the MOVED meshes' leaf boxes against what the build gave them (1 after a build; chained over the
                                  refits since; two-level scenes: the worst BLAS).  The sharper of the two signals: a small, finely tessellated
                                  mesh in a large room hardly moves the whole tree's cost (DragonScene at a 2 % deformation: wide_cost x 1.014,
                                  leaf_growth ~2, rate x 0.85).  "refit_max_cost_ratio" acts on whichever is larger.  Both are measured on the
                                  8-wide layout: a scene built without it (wide = 0) refits its rope layout and reports 0 / 1 here       */
} MRTSceneStats;

typedef struct {
    uint64_t frames;           /* frames rendered since create/resize                          */
    uint64_t closest_rays;     /* R_closest summed over those frames                           */
    uint64_t shadow_rays;      /* R_shadow (shadow rays actually cast, Raytracing.metal:341)   */
    uint64_t primary_rays;     /* w*h (this shard's pixels) per frame, summed                  */
    uint64_t bytes_alg;        /* SURVEY §8(d) algorithmic bytes, summed                       */
    float    ms_gpu_last;      /* device ms of the last mrt_renderer_render batch (HIP events) */
    float    ms_extend_last;   /* device ms spent in the closest-hit kernel within that batch  */
    uint32_t extend_launches_last;
    uint32_t _pad;
} MRTRenderStats;

/* Device time per kernel class over the launches of the last mrt_renderer_render call that carried their own start/stop
 * events (the first 512 launches; hipExtLaunchKernelGGL events, the clock rocprofv3 --kernel-trace reads).               */
enum { MRT_KERNEL_PRIMARY = 0,   /* primary-ray generation + first closest hit                      */
       MRT_KERNEL_SHADE = 1,     /* normals, light sampling, NEE / bounce ray emission, compaction   */
       MRT_KERNEL_TRACE = 2,     /* bounce rays (closest hit) + shadow rays (any hit): the dominant kernel */
       MRT_KERNEL_ACCUMULATE = 3,
       MRT_KERNEL_CLASSES = 4 };
typedef struct {
    float    ms[MRT_KERNEL_CLASSES];          /* summed duration of the timed launches of the class   */
    uint32_t launches[MRT_KERNEL_CLASSES];    /* how many launches that sum covers                     */
} MRTKernelTimes;

typedef struct MRTContext_  *MRTContext;
typedef struct MRTScene_    *MRTScene;
typedef struct MRTRenderer_ *MRTRenderer;
typedef struct MRTGroup_         *MRTGroup;            /* n devices of one node driven by one process               */
typedef struct MRTGroupRenderer_ *MRTGroupRenderer;    /* one renderer per device of a group, image sharded by tile  */

/* ---------------------------------------------------------------- errors */
/* Message of the last failing call on this thread ("" if none).                               */
const char *mrt_last_error(void);
int         mrt_abi_version(void);

/* ---------------------------------------------------------------- context
 * replaces MTLCreateSystemDefaultDevice + makeCommandQueue (Renderer.swift:46-59).
 * Fails with MRT_ERR_NO_DEVICE when there is no HIP device — there is no CPU fallback.         */
int mrt_context_create(int device_id, MRTContext *out);
/* Handles are not reference-counted (the reference's objects are, by ARC): destroy renderers before their scene, and scenes and
 * renderers before their context.  Out of order the call is refused — MRT_ERR_STATE, nothing is freed — instead of leaving a handle
 * that points at freed memory.  NULL is accepted everywhere.                                                                    */
int mrt_context_destroy(MRTContext ctx);
/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int mrt_context_set_stream(MRTContext ctx, void *hip_stream);
int mrt_context_device_name(MRTContext ctx, char *buf, size_t buflen);

/* ---------------------------------------------------------------- scene / geometry
 * replaces Model.init / Mesh.init / Submesh.init (Model.swift:13-24, Mesh.swift:18-33,
 * SubMesh.swift:23-33).  The library copies caller arrays (as MTKMeshBufferAllocator does).   */
int mrt_scene_create(MRTContext ctx, MRTScene *out);
int mrt_scene_destroy(MRTScene scene);

/* One Mesh = one instance (Renderer.swift:193-200).  positions/normals: nverts vectors with the
 * given byte stride (16 for the reference's float3 buffers, 12 for packed).  transform: 16
 * floats, column-major 4x4 object→world (Mesh.swift:21-24); the last row is ignored exactly as
 * matrix4x4_drop_last_row does (Utilities.swift:92-101).  Returns the instance id in *mesh_id.
 * Positions, normals and transforms must be finite: a NaN or an infinity is MRT_ERR_INVALID_ARGUMENT
 * here, in mrt_scene_add_instance, mrt_scene_update_mesh and mrt_scene_set_instance_transform (the
 * scene keeps what it had).                                                                     */
int mrt_scene_add_mesh(MRTScene scene, const float *positions, size_t pos_stride_bytes,
                       const float *normals, size_t nrm_stride_bytes, size_t nverts,
                       const float *transform_colmajor_4x4, int32_t *mesh_id);
/* One Submesh = one geometry of that mesh's primitive AS (Mesh.swift:39-48).  indices: 3*ntris
 * uint32 into the mesh's vertex arrays.  Returns the geometry id in *geometry_id.              */
/* A further instance of an existing mesh: shares its vertex arrays, submeshes and materials, has its own transform and its own
 * mesh id (= instance id).  The reference's counterpart is an MTLAccelerationStructureInstanceDescriptor whose
 * accelerationStructureIndex names an existing primitive structure (Renderer.swift:193-203).  With scene option instancing = 1
 * the geometry gets ONE bottom-level BVH shared by all its instances; with instancing = 0 (default) instances are flattened.   */
int mrt_scene_add_instance(MRTScene scene, int32_t source_mesh_id, const float *transform4x4_colmajor, int32_t *mesh_id);
int mrt_mesh_add_submesh(MRTScene scene, int32_t mesh_id, const uint32_t *indices, size_t ntris,
                         const MRTMaterial *material, int32_t *geometry_id);
/* Convenience = Model(name:position:rotation:scale:) (Model.swift:13): read OBJ+MTL with the
 * library's reader, build T*R*S (Mesh.swift:21-24, Utilities.swift:113-166), add mesh+submeshes. */
int mrt_scene_add_obj(MRTScene scene, const char *obj_path, const float position[3],
                      const float rotation[3], float scale, int32_t *mesh_id);
/* Scene.lights → lightBuffer (Scene.swift:15-33).                                              */
int mrt_scene_set_lights(MRTScene scene, const MRTLight *lights, int32_t count);
/* createAccelerationStructures (Renderer.swift:184-214): on-device BVH build; blocking, like the
 * reference's waitUntilCompleted (Utilities.swift:63,83).  builder: 0 = default.               */
int mrt_scene_commit(MRTScene scene);
int mrt_scene_set_option(MRTScene scene, const char *key, double value);
/* Animated transforms: replace one instance's object->world matrix (MTLAccelerationStructureInstanceDescriptor
 * .transformationMatrix, Renderer.swift:193-200); takes effect at the next mrt_scene_commit, which rebuilds the
 * world-space BVH on the device (the reference would refit/rebuild its instance AS, Renderer.swift:205-213).   */
int mrt_scene_set_instance_transform(MRTScene scene, int32_t mesh_id, const float *transform_colmajor_4x4);
/* Deforming geometry: new object-space positions and normals for the vertices of one mesh (same count, same submesh indices: the topology is kept).  The library copies
 * the arrays.  Takes effect at the next mrt_scene_commit.  When nothing else changed since the last commit, a scene with the 8-wide layout REFITS — a flattened scene its tree,
 * a two-level scene (instancing = 1) the BLAS of every changed mesh in place, in both layouts, and then its TLAS: every triangle packet rewritten, the boxes recomputed
 * bottom-up, the tree's shape as built — in a fraction of a build's time; the image is the one a fresh build of the deformed scene gives (the closest hit does not depend on
 * the tree).  The boxes of a tree that keeps its shape loosen with the deformation: MRTSceneStats.wide_cost against wide_cost_built says by how much (sah_cost follows), and
 * scene option "refit_max_cost_ratio" = r makes a commit build again by itself once wide_cost > r x wide_cost_built.  A rope layout (scene option rope = 1 beside the 8-wide one, or wide = 0 alone) is
 * refitted the same way; scene option refit = 0 builds again.  The reference builds its acceleration structures once (Renderer.swift:184-214) and never deforms a mesh; this is the counterpart of Metal's refit
 * of a primitive acceleration structure.                                                                                                                              */
int mrt_scene_update_mesh(MRTScene scene, int32_t mesh_id, const float *positions, size_t pos_stride_bytes,
                          const float *normals, size_t nrm_stride_bytes, size_t vertex_count);
int mrt_scene_stats(MRTScene scene, MRTSceneStats *out);
/* 4x3 packed instance transform as the reference stores it (Renderer.swift:193-203).          */
int mrt_scene_instance_transform(MRTScene scene, int32_t mesh_id, float out_colmajor_4x3[12]);

/* Intersector queries against the committed scene: the two uses of `intersector.intersect`
 * (Raytracing.metal:244 closest, :367 any).  Host arrays in, host arrays out.                  */
int mrt_scene_intersect_closest(MRTScene scene, const MRTRay *rays, size_t n, MRTIntersection *out);
int mrt_scene_intersect_any(MRTScene scene, const MRTRay *rays, size_t n, int32_t *occluded);

/* ---------------------------------------------------------------- host-side geometry helpers
 * (no GPU needed) — the library's OBJ/MTL reader standing in for ModelIO (Model.swift:16-21,
 * SubMesh.swift:37-54) and the procedural dragon proxy (dragon.obj is absent upstream).        */
typedef struct MRTMeshData_ *MRTMeshData;
int mrt_obj_load(const char *obj_path, MRTMeshData *out);
int mrt_dragon_proxy(MRTMeshData *out);             /* exactly 871 414 triangles               */
int mrt_dragon_proxy_irregular(MRTMeshData *out);   /* same count / extents / material, irregular connectivity, shuffled order (sensitivity check) */
int mrt_dragon_proxy_hostile(MRTMeshData *out);     /* same count / extents / material; triangle sizes over 100 : 1 and 1 % slivers of up to 50 x their edge (stress test of the builder) */
int mrt_bunny_proxy(MRTMeshData *out);              /* exactly  69 451 triangles               */
int mrt_meshdata_free(MRTMeshData m);
int mrt_meshdata_counts(MRTMeshData m, size_t *nverts, int32_t *nsubmeshes);
/* positions / normals: nverts*3 packed floats */
int mrt_meshdata_vertices(MRTMeshData m, float *positions, float *normals);
int mrt_meshdata_submesh(MRTMeshData m, int32_t submesh, size_t *ntris, uint32_t *indices /* may be NULL */,
                         MRTMaterial *material /* may be NULL */, char *name_buf, size_t name_buflen);
/* T*R*S with R = Rx*Ry*Rz (Mesh.swift:21-24; Utilities.swift:104-166), column-major 4x4.       */
int mrt_make_transform(const float position[3], const float rotation[3], float scale, float out16[16]);
/* Scene.setupCamera(size:) (Scene.swift:40-57).                                                */
int mrt_default_camera(int32_t width, int32_t height, MRTCamera *out);

/* ---------------------------------------------------------------- renderer
 * replaces Renderer.init / createTextures / createBuffers (Renderer.swift:45-71, :107-182,
 * :231-275).  seed drives the per-pixel Halton offsets (the reference uses arc4random, :259).
 * max_bounces: the literal 3 of Raytracing.metal:237.                                         */
int mrt_renderer_create(MRTContext ctx, MRTScene scene, int32_t width, int32_t height,
                        uint32_t seed, int32_t max_bounces, MRTRenderer *out);
int mrt_renderer_destroy(MRTRenderer r);
/* mtkView(_:drawableSizeWillChange:) (Renderer.swift:353-356): new targets, new seeds, frameIndex=0 */
int mrt_renderer_resize(MRTRenderer r, int32_t width, int32_t height);
/* Default camera = Scene.setupCamera(size) recomputed from the size (Scene.swift:36-57).       */
int mrt_renderer_set_camera(MRTRenderer r, const MRTCamera *camera);
/* updateUniforms (Renderer.swift:216-229) in one call: the 96-byte Uniforms block the reference binds at buffer index 0
 * (ShaderTypes.h:89-97, Renderer.swift:304).  A new width / height resizes (new targets and seeds, Renderer.swift:353-356, and
 * the given frameIndex is applied after that); frameIndex is the accumulation weight and Halton index of the next frame;
 * lightCount in [1, lights of the scene] makes the kernels sample only the first lightCount lights (Raytracing.metal:273, :335);
 * blocksWide is ignored (derived from the size; the reference's kernel never reads it).  get_uniforms returns what the next
 * frame will be drawn with.                                                                                                 */
int mrt_renderer_set_uniforms(MRTRenderer r, const MRTUniforms *uniforms);
int mrt_renderer_get_uniforms(MRTRenderer r, MRTUniforms *uniforms);
/* The renderer's knobs.  The reference's own: "max_bounces" (the literal 3 of Raytracing.metal:237; 1..19), "frames_in_flight"
 * (Renderer.maxFramesInFlight, Renderer.swift:33: here passes in flight on separate HIP streams, default 3 as the reference), "sample_offset" (added to
 * frameIndex for the Halton index only: sample-index sharding).  This implementation's: "frame_batch" (frames carried through the pipeline
 * per pass, 1..32: larger launches against more queue memory — "lane_bytes" per pass in flight; 0, the default, sizes it by the image: 8 at 1920 x 1080 pixels
 * per device and above, proportionally more for a smaller image or a shard of one, so that a pass always carries about the same number of pixel-frames; reads back as the value in force), "megakernel" (1: one launch
 * per frame, the lowest latency of a single frame; the default pipeline has the higher throughput), "materials" (1: the materials
 * extension — emission, specular lobe, refraction; max_bounces <= 16; the only key that changes the image).  Read-only through
 * mrt_renderer_get_option: "lanes_used", "lane_bytes".                                                                              */
int mrt_renderer_set_option(MRTRenderer r, const char *key, double value);
int mrt_renderer_get_option(MRTRenderer r, const char *key, double *value);
/* Screen-tile shard for multi-GPU: this renderer owns 8x8 tiles with (tile_id % world) == rank;
 * other pixels stay 0 in its targets so that a sum-reduce assembles the frame.                 */
int mrt_renderer_set_shard(MRTRenderer r, int32_t rank, int32_t world);
/* Restart accumulation at a given frame index (sample-index sharding; resume).                 */
int mrt_renderer_set_frame_index(MRTRenderer r, uint32_t frame_index);
int mrt_renderer_frame_index(MRTRenderer r, uint32_t *frame_index);
/* draw(in:) (Renderer.swift:284-351) n_frames times: enqueue on the stream and return.         */
int mrt_renderer_render(MRTRenderer r, int32_t n_frames);
/* commandBuffer completion (Renderer.swift:285-287).                                           */
int mrt_renderer_wait(MRTRenderer r);
/* The same completion as a poll (the reference's handler is told per command buffer, Renderer.swift:285-287): frames, counted
 * as MRTRenderStats.frames is, whose accumulation has finished on the device.  Never blocks.  Granularity: a pass of `frame_batch`
 * frames; the LAST passes of a draw call (one per pass in flight, i.e. every pass of a short call such as 20 frames) are accumulated
 * together when the call's last traversal launch has finished, so their frames are reported together at the end of the call.  A call
 * whose passes carry ONE frame each runs every pass as groups of tiles on several streams and reports its frames at the end of the call.  */
int mrt_renderer_frames_completed(MRTRenderer r, uint64_t *frames);
/* accumulationTargets[0] after the swap (Renderer.swift:332-334): w*h RGBA32F, row 0 = bottom of
 * the image as the kernel writes it (Raytracing.metal:206-207; the blit flips, Shaders.metal:35). */
int mrt_renderer_read_accum(MRTRenderer r, float *rgba, size_t nbytes);
/* Same data copied device→device into caller memory (e.g. a torch tensor for the RCCL reduce). */
int mrt_renderer_copy_accum_to_device(MRTRenderer r, void *device_ptr, size_t nbytes);
int mrt_renderer_write_accum_from_device(MRTRenderer r, const void *device_ptr, size_t nbytes);
/* The compact assemble of a tile-sharded image (beside the reduce(sum) of whole buffers that BASELINE.json's north_star prescribes and mrt_group_gather defaults to): a rank
 * ships only the pixels it owns — 1 / world of the image — and the root writes them in place.  A compact buffer is tiles x 64 RGBA32F pixels, device memory: tile lt of shard
 * (rank, world) is tile lt * world + rank of the image (8 x 8 tiles, row-major over the image), its pixels row-major, pixels outside the image 0.
 *   _shard_tiles   tiles of this renderer's image that shard (rank, world) owns
 *   _pack_owned    this renderer's accumulation buffer -> the compact buffer of ITS shard (mrt_renderer_set_shard), enqueued on its stream
 *   _unpack_tiles  the compact buffer of shard (rank, world) -> this renderer's accumulation buffer, at those tiles' pixels                                          */
int mrt_renderer_shard_tiles(MRTRenderer r, int32_t rank, int32_t world, uint64_t *tiles);
int mrt_renderer_pack_owned_tiles(MRTRenderer r, void *device_ptr, size_t nbytes);
int mrt_renderer_unpack_tiles(MRTRenderer r, const void *device_ptr, size_t nbytes, int32_t rank, int32_t world);
/* fragmentShader (Shaders.metal:39-52): Reinhard c/(1+c), top row first (flipped), RGBA8.      */
int mrt_renderer_read_tonemapped_rgba8(MRTRenderer r, uint8_t *rgba, size_t nbytes);
int mrt_renderer_stats(MRTRenderer r, MRTRenderStats *out);
int mrt_renderer_reset_stats(MRTRenderer r);
int mrt_renderer_kernel_times(MRTRenderer r, MRTKernelTimes *out);   /* waits for the last render call */

/* ---------------------------------------------------------------- device group (multi-GPU, one process)
 * The reference creates ONE device and ONE queue (MTLCreateSystemDefaultDevice + makeCommandQueue, Renderer.swift:46-59); this is that
 * seam widened to the n GPUs of a node.  The scene and its BVH are replicated on every device; the image is sharded by 8x8 screen tile
 * (tile_id % n == rank); each device accumulates its frames locally; mrt_group_gather assembles the image with ONE reduce(sum) of the
 * RGBA32F buffer — ncclReduce over xGMI (RCCL is opened when a group of several distinct devices is created), or peer copies + add.
 * The assembled image is bit-identical to the single-device image.                                                                  */
int mrt_group_create(const int *device_ids, int32_t n, MRTGroup *out);
int mrt_group_destroy(MRTGroup g);
int mrt_group_size(MRTGroup g, int32_t *n);
int mrt_group_context(MRTGroup g, int32_t rank, MRTContext *ctx);                  /* borrowed: the group owns its contexts            */
/* mode: 0 = ncclReduce(sum, float32, root 0) of the whole buffers, 1 = peer copies of the whole buffers into the root device + add, 2 = compact: every rank packs the tiles it
 * owns (1 / n of the image) and the root receives them (ncclSend / ncclRecv, or peer copies) and writes them in place; note: why (may be NULL).  Same image bit for bit.        */
int mrt_group_reduce_mode(MRTGroup g, int32_t *mode, char *note, size_t note_len);
int mrt_group_set_reduce_mode(MRTGroup g, int32_t mode);
/* Renderer.init for the whole group: `scene` (any context; committed or not) is the template — its meshes, lights and build options are
 * replicated and committed on every device; rank r renders the tiles with tile_id % n == r.  The template scene stays the caller's.  */
int mrt_group_renderer_create(MRTGroup g, MRTScene scene, int32_t width, int32_t height, uint32_t seed, int32_t max_bounces, MRTGroupRenderer *out);
int mrt_group_renderer_destroy(MRTGroupRenderer gr);
int mrt_group_renderer_rank(MRTGroupRenderer gr, int32_t rank, MRTRenderer *r);     /* borrowed: one device's renderer (options, stats)  */
int mrt_group_set_option(MRTGroupRenderer gr, const char *key, double value);      /* mrt_renderer_set_option on every device           */
int mrt_group_set_camera(MRTGroupRenderer gr, const MRTCamera *camera);
/* draw(in:) (Renderer.swift:284-351) n_frames times on every device: enqueues and returns; the devices run concurrently.              */
int mrt_group_render(MRTGroupRenderer gr, int32_t n_frames);
int mrt_group_wait(MRTGroupRenderer gr);
int mrt_group_frames_completed(MRTGroupRenderer gr, uint64_t *frames);             /* minimum over the devices; never blocks            */
/* The one collective per output image: reduce(sum) of every device's accumulation buffer into rank 0, then (rgba != NULL) a copy of
 * the assembled w*h RGBA32F image to the host (row 0 = bottom, as mrt_renderer_read_accum).  Blocks until the image is assembled.   */
int mrt_group_gather(MRTGroupRenderer gr, float *rgba, size_t nbytes);
int mrt_group_gathered_device_ptr(MRTGroupRenderer gr, void **device_ptr);         /* the assembled image on the root device            */
int mrt_group_stats(MRTGroupRenderer gr, MRTRenderStats *out);                     /* ray counters summed over the devices              */

/* Diagnostics, device-function probes and the library-internal A/B switches (mrt_debug_*) are declared in mrt_debug.h — used by tests/,
 * tools/ and bench.py, not part of the host contract and not installed with this header.                                              */

#ifdef __cplusplus
}
#endif
#endif /* MRT_ABI_H */
