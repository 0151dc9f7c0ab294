// renderer.h — host frame driver state (Renderer.swift:12-43 restated for HIP).
#pragma once
#include "scene_device.h"
#include "host_geometry.h"
#include <array>
#include <deque>
#include <vector>

namespace mrt {

struct EvPair { hipEvent_t a = nullptr, b = nullptr; int kind = 0; };    // kind: MRT_KERNEL_* (mrt_abi.h)

// One pass in flight (the reference keeps 3 frames in flight, Renderer.maxFramesInFlight, Renderer.swift:33): its
// own HIP stream, ray / hit / shadow queues, per-pixel sample buffers and per-bounce queue counters, all sized for
// a batch of `frame_batch` consecutive frames.  Passes on different lanes overlap on the GPU (one pass's straggler
// waves are covered by the next pass's bulk); only the accumulate step is ordered pass to pass, by an event.
struct FrameLane {
    hipStream_t stream = nullptr;
    hipEvent_t accumulated = nullptr;    // recorded after this lane's k_accumulate
    DevBuf<float4> sample;               // [sub-frame][pixel] radiance of the batch's frames
    DevBuf<float4> rayA[2], rayB[2], thr[2], hits, srayA, srayB, scon;
    DevBuf<unsigned long long> bounce_counts;   // per bounce {next-queue rays (lo 32), shadow rays (hi 32)}
    // shadow planes (default path): the light's contribution of bounces 1 and 2 per sample (bounce 0 uses `sample`) and one byte per sample and bounce "the shadow ray got through"
    DevBuf<float4> f_con[2];
    DevBuf<uint8_t> f_lit;               // [sub-frame][pixel][bounce]: one 32-bit word per sample
    DevBuf<uint4> pairs;                 // two-level scenes, binned walk (tl_pairs): {ray, instance, bound, report tag} queued by the TLAS pass for the BLAS pass; allocated at the first such pass
};
constexpr int MAX_FRAMES_IN_FLIGHT = 16;
constexpr int MAX_TILE_GROUPS = 4;
// One tile group of a pass: the renderer's own shard (rank of world) cut once more by local tile index — group g of G takes the shard's tiles g, g + G, ... — which is
// exactly shard (g * world + rank) of (G * world).  A short draw (fewer passes than lanes) runs every pass as G such groups on G lanes: the groups are disjoint sets of
// pixels writing the same accumulation target, so one group's shade launches overlap another's traversal launches and no launch waits for another group (Renderer::tile_groups).
struct TileGroup { int rank = 0, world = 1, tiles_local = 0; uint32_t capacity = 0; uint64_t owned = 0; const uint32_t *seeds = nullptr; };
constexpr int MAX_FRAME_BATCH = 32;
constexpr int DEFAULT_FRAME_BATCH = 8;

struct Renderer {
    hipStream_t stream = nullptr;
    const DeviceScene *scene = nullptr;
    int width = 0, height = 0, max_bounces = 3;
    uint32_t seed = 1;
    MRTCamera camera{};
    uint32_t frame_index = 0;            // Renderer.frameIndex (Renderer.swift:41)
    uint32_t sample_offset = 0;          // added to frame_index for the Halton index only (sample-index sharding)
    int cur = 0;                         // accumulationTargets[0] == accum[cur] after the swap (:332-334)
    int shard_rank = 0, shard_world = 1;
    int tiles_local = 0;
    uint32_t capacity = 0;               // ray-queue capacity = local tiles * 64
    uint64_t owned_pixels = 0;
    uint64_t frames_rendered = 0;

    DevBuf<uint32_t> seeds;              // randomTexture (R32Uint, :246-274)
    DevBuf<float> halton_tab;            // bounce 0's Halton values by index, dimensions 1 .. 6 (renderer.hip FrameParams::halton_tab), for the window [halton_w0, halton_w0 + 2^20 + 2^16); allocated by the first bundled pass
    uint32_t halton_w0 = 0;
    int halton_table = 1;                // renderer option: bundled passes read bounce 0's Halton values from the table
    int ensure_halton_table(uint32_t sample_index, uint32_t frames);          // once per draw, on the main stream ahead of the fork
    DevBuf<uint32_t> hint;               // per pixel: the packet its primary ray hit last (k_trace_primary tests it first); 0xFFFFFFFF = none
    bool primary_hint = true;
    bool throughput_chain = true;        // bounce rays carry the resource slots of their path instead of a throughput record (renderer.hip FrameParams::chain)
    DevBuf<float4> accum[2];             // accumulationTargets (RGBA32F, :231-244)
    FrameLane lanes[MAX_FRAMES_IN_FLIGHT];
    int frames_in_flight = 3;            // Renderer.maxFramesInFlight is 3 (Renderer.swift:33), and three 8-frame passes in flight reach the rate of six on the round-6 tree (14.1 = 14.1 Grays/s at 240 steps, 12.5 = 12.5 at 20;
                                         // profiles/r06_lanes.txt) with half the queue memory: 8.2 instead of 16.3 GB at 1080p.  Tile groups of one-frame passes take the lanes they need beyond this
    int frame_batch = 0;                 // frames carried through the pipeline per pass at most (a draw's frames go in passes of equal size); 1 = one frame per pass; 0 (default) = by image size:
                                         // DEFAULT_FRAME_BATCH at 1920 x 1080 pixels per device and above, proportionally more for a smaller image or a shard of one (batch_wanted())
    int batch_wanted() const;            // the option, or what "by image size" comes to for this renderer's pixels
    int tile_groups = 0;                 // a pass as G tile groups on G lanes: 0 = by the draw (G > 1 only when it has fewer passes than lanes), 1 = never, 2..4 = that many whenever the lanes are there
    int groups_used = 1;                 // G of the last draw
    TileGroup tgroups[MAX_TILE_GROUPS];  // valid for tgroups_for groups at batch tgroups_batch (0 = not built)
    DevBuf<uint32_t> seeds_g[MAX_TILE_GROUPS]; int tgroups_for = 0, tgroups_batch = 0;
    int ensure_tile_groups(int G);
    int lanes_used = 0;                  // lanes the last draw ran on (<= frames_in_flight when device memory is short)
    int lanes_ready = 0;                 // lanes [0, lanes_ready) hold queues and sample buffers
    int alloc_batch = 0;                 // batch the queues / sample buffers / seed table are sized for
    bool megakernel = false;             // one launch per frame (k_megakernel): lowest latency of a single frame; the wavefront pipeline has the higher throughput
    int mega_slots = 0; size_t mega_slots_for_stack = ~(size_t)0;
    bool materials = false;              // the materials extension: emission, specular lobe, dielectric refraction (shade_entry<MATERIALS>); off = the reference's diffuse-only kernel
    int primary_wide = 2;                // primary rays of a flattened scene: 2 = one ray per lane on the 8-wide layout (inside shade(0) or in their own launch; default), 1 = the 8-wide stream kernel with lane refill (own launch), 0 = the rope walk (scene option rope = 1)
    int persistent = 2;                  // bounce / shadow traversal as persistent waves pulling chunks of rays from a shared counter: 0 never, 1 always, 2 by launch size
    int xcd_counters = 1;                  // pulling traversal launches: 1 = one work counter and one eighth of every sub-frame's rays per XCD (traverse_wide.h XcdRegions), 0 = one counter for all
    int hit_lds = 1;                     // pulling traversal launches of flattened scenes: a lane's closest hit keeps U, V, |det| and id in LDS; a finished ray is reported without re-testing its triangle (traverse_wide.h StreamExt)
    int shade_pack = 1;                  // k_shade of bounces >= 1 compacts the hits of its queue in LDS and shades them on full waves (k_shade_pack)
    int wave_slots_x = 0; int slots_x_key = -1;      // wave slots of that kernel (k_trace_mixed_wide_persist_x), and the LDS size they were computed for
    int persist_chunk = 256;             // rays per pull (upper bound; small queues pull less, see render())
    int wave_slots = 7168;               // resident waves the persistent launch is sized for (occupancy query at the first draw)
    bool wave_slots_user = false;        // set through the option: keep it
    size_t slots_for_stack = ~(size_t)0;
    int alloc_planes(FrameLane &L);
    bool tail_accumulate = true;         // the last passes of a draw (one per lane) are accumulated in one launch after the join instead of one after the other
    int fuse_primary = 1;                // the primary rays are generated, traced and shaded in ONE launch (k_shade_primary): no hit / direction records, one launch less per pass
    int shadow_planes = 1;               // the light's contribution per pixel and bounce + one byte per shadow ray that got through, instead of a contribution queue and a read-modify-write of the sample buffer (renderer.hip k_accumulate_planes)
    int equal_passes = 1;                // a draw's frames go in passes of equal size (20 frames at frame_batch 8: 7 + 7 + 6); 0: full passes first (8 + 8 + 4) — measured worse
    int frame_bundle = 1;                // bounce 0 of a multi-frame pass: a wave of k_shade_primary takes 8 slots x 8 sub-frames (FrameParams::frame_bundle)
    int stream_stride = 2;               // the static split deals 64-ray batches round-robin to the waves (BatchStride) instead of one contiguous range each: 0 never, 1 always, 2 = a shard's launches (with one round of waves)
    int stream_even = 200;               // a traversal launch too small for chunk pulling has stream_even % of the wave slots as waves and splits the rays its queue really holds evenly among them (k_trace_mixed_wide_stream); 0 = rays_per_wave each, grid sized for the queue's capacity
    int tl_pair_cap = 0;                 // test aid: > 0 bounds the pair queue (pushes beyond it walk their instance in place); 0 = one pair per virtual ray
                                         // Measured on dragon x 4 (profiles/r06_two_level_ab.txt): 7.4 against 9.2 Grays/s — the pass is ~225 M wave-instructions per 8-frame launch, 0.55 ms of VALU issue wherever it runs; the shade kernels' idle VALU time (0.2 ms per launch) cannot hide it
    int tl_pairs = 1;                    // two-level scenes: bounce / shadow rays as TLAS pass + BLAS pass over (ray, instance) pairs (k_tl_top / k_tl_blas) instead of one loop over both levels; 0 = the one-loop walk
    bool wide_bounce = true;             // A/B switch: 0 = bounce / shadow rays on the rope kernels although the scene has the 8-wide layout
    hipEvent_t ev_fork = nullptr;
    DevBuf<unsigned long long> totals;   // [0] closest rays, [1] shadow rays, [2] primary rays

    int light_count_limit = 0;           // > 0: the kernels see only the first n lights (Uniforms.lightCount, ShaderTypes.h:93; 0 = all the scene's lights)
    // completion without blocking (the reference is told per frame, Renderer.swift:285-287): one pooled event per pass, recorded after its
    // k_accumulate; passes complete in order (each accumulate waits for the previous one)
    struct PassDone { hipEvent_t ev; uint64_t frames_through; };
    std::deque<PassDone> passes_pending;
    std::vector<hipEvent_t> pass_events_free;
    uint64_t frames_completed_known = 0; // frames (since create / resize) whose accumulate is known to have finished
    int note_pass(hipStream_t st);       // render(): after a pass's accumulate
    int poll_completed(uint64_t *out);   // hipEventQuery only; never waits

    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::array<EvPair, 512> ev_ext;
    int ext_used = 0;
    bool pending_timing = false;
    float ms_last = 0, ms_extend_last = 0; uint32_t extend_launches_last = 0;
    MRTKernelTimes kernel_times{};       // per kernel class, over the launches of the last render() that carried events

    Renderer() = default;
    Renderer(const Renderer &) = delete;
    ~Renderer();
    int init(hipStream_t st, const DeviceScene *sc, int w, int h, uint32_t seed, int max_bounces);
    int resize(int w, int h);
    int alloc_queues();
    int alloc_lane(FrameLane &L);
    void release_lane(FrameLane &L);
    size_t lane_bytes() const;           // device bytes of one lane's queues + sample buffer at the current size and batch
    int set_shard(int rank, int world);
    int render(int n_frames);
    int wait();
    int read_accum(float *rgba, size_t nbytes);
    int copy_accum_to_device(void *dptr, size_t nbytes);
    int write_accum_from_device(const void *dptr, size_t nbytes);
    int read_tonemapped(uint8_t *rgba, size_t nbytes);
    int pack_owned_tiles(void *dptr, size_t nbytes);                                   // accum -> compact [tiles_local][64] buffer of this renderer's shard (device pointer; on `stream`)
    int unpack_tiles(const void *dptr, size_t nbytes, int rank, int world);            // compact buffer of shard (rank, world) -> accum at those tiles' pixels
    int stats(MRTRenderStats *out);
    int reset_stats();
};

uint32_t shard_tiles(int width, int height, int rank, int world);                     // 8 x 8 tiles of the image that shard (rank, world) owns: tile_id % world == rank
int unpack_tiles_into(float4 *image, int width, int height, const void *compact, size_t nbytes, int rank, int world, hipStream_t st);
int query_closest(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, MRTIntersection *out);
int query_any(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, int32_t *out);
int query_stats(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, int any, uint32_t *out4);
int query_stream(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, int any, MRTIntersection *out);
int query_stream_stats(const DeviceScene &sc, hipStream_t stream, const MRTRay *rays, size_t n, int any, uint32_t per_wave, uint32_t *out8, size_t nwaves);
int probe_halton(hipStream_t stream, const int32_t *i, const int32_t *d, size_t n, float *out);
int probe_hemisphere(hipStream_t stream, const float *u2, const float *n3, size_t n, float *out3);
int probe_seeds(hipStream_t stream, uint32_t seed, int w, int h, uint32_t *out);
int calibrate(hipStream_t stream, size_t table_bytes, double *out3);   // calibrate.hip

}  // namespace mrt
