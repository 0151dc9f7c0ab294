// group.hip — one process, n devices (include/mrt_abi.h "device group"): the multi-GPU seam of the reference's single-device
// Renderer (Renderer.swift:46-59 creates ONE MTLDevice and ONE command queue; SURVEY §8b asks for ctx_create(device_ids[], n)).
//
// The scene and its BVH are REPLICATED on every device of the group (a few hundred MB against 288 GB of HBM); the image is sharded by 8x8
// screen tile, tile_id % n == rank (Renderer::set_shard), so that expensive and cheap regions of the image interleave across the devices;
// every device accumulates its frames locally into a zero-initialised full-frame RGBA32F buffer; mrt_group_gather assembles the image
// with ONE reduce(sum) of that buffer per output image — ncclReduce over xGMI (RCCL, opened with dlopen when the first group of more than
// one distinct device is created: a process that already holds an RCCL, e.g. PyTorch's, shares it), or, selectable, peer copies into the
// root device + an add kernel (also what a group that names one device several times uses: RCCL refuses duplicate devices).
// Shards are disjoint and every other pixel of a shard's buffer is +0, so the sum is exact: the assembled image is bit-identical to the
// single-device image (tests/test_group.py).
#include "api_types.h"
#include <rccl/rccl.h>          // types and enums only: the entry points are resolved with dlsym
#include <dlfcn.h>
#include <cstring>
#include <cstdio>
#include <memory>
#include <algorithm>
#include <set>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;          // (optional: the compact assemble falls back to peer copies without them)
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    bool load() {
        if (handle) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { why = std::string("librccl.so.1 could not be opened: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        Reduce = (decltype(Reduce))dlsym(handle, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        Send = (decltype(Send))dlsym(handle, "ncclSend"); Recv = (decltype(Recv))dlsym(handle, "ncclRecv");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !GetErrorString) { why = "librccl.so.1 lacks an entry point"; handle = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;

int rccl_fail(ncclResult_t e, const char *what) {
    mrt::set_error(std::string("RCCL error in ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?"));
    return MRT_ERR_HIP;
}
#define MRT_RCCL(call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return rccl_fail(e_, #call); } while (0)

__global__ void k_add4(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float4 a = dst[i], b = src[i]; dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
}

// One host thread per device of a group (rank 0 = the caller's own thread): committing the scene (the BVH build blocks), creating the renderer and enqueueing a draw
// are host work per device — 8.4 ms, a few ms and 0.2-0.4 ms for a 20-frame draw — which one thread would do for eight devices one after the other while
// a rank of eight has 2.5 ms of GPU work.  run(fn) calls fn(rank) for every rank at once and returns the first failure (its message becomes the caller's mrt_last_error).
// The threads live as long as the group; if they cannot be created the ranks simply run one after the other on the caller's thread.
class RankPool {
    struct Worker {
        std::thread th; std::mutex m; std::condition_variable cv;
        std::function<int()> job; bool has_job = false, done = false, quit = false; int rc = MRT_OK; std::string err;
    };
    std::vector<std::unique_ptr<Worker>> w;
    static void loop(Worker *k) {
        for (;;) {
            std::function<int()> job;
            { std::unique_lock<std::mutex> l(k->m); k->cv.wait(l, [&] { return k->has_job || k->quit; }); if (k->quit) return; job = std::move(k->job); k->has_job = false; }
            int rc; std::string err;
            try { rc = job(); if (rc) err = mrt_last_error(); }
            catch (const std::exception &e) { rc = MRT_ERR_INVALID_ARGUMENT; err = std::string("exception: ") + e.what(); }
            catch (...) { rc = MRT_ERR_INVALID_ARGUMENT; err = "unknown exception"; }
            { std::lock_guard<std::mutex> l(k->m); k->rc = rc; k->err = std::move(err); k->done = true; }
            k->cv.notify_all();
        }
    }
public:
    RankPool() = default;
    RankPool(const RankPool &) = delete; RankPool &operator=(const RankPool &) = delete;
    void start(int ranks) {
        for (int r = 1; r < ranks; r++) {
            std::unique_ptr<Worker> k(new Worker());
            try { k->th = std::thread(loop, k.get()); } catch (...) { break; }       // no more threads to be had: the remaining ranks run on the caller's thread
            w.push_back(std::move(k));
        }
    }
    ~RankPool() {
        for (auto &k : w) { { std::lock_guard<std::mutex> l(k->m); k->quit = true; } k->cv.notify_all(); }
        for (auto &k : w) if (k->th.joinable()) k->th.join();
    }
    int run(int ranks, const std::function<int(int)> &fn) {
        const int par = std::min(ranks - 1, (int)w.size());
        for (int r = 1; r <= par; r++) { Worker *k = w[(size_t)r - 1].get(); { std::lock_guard<std::mutex> l(k->m); k->job = [&fn, r] { return fn(r); }; k->has_job = true; k->done = false; } k->cv.notify_all(); }
        int rc = MRT_OK; std::string err;
        auto inline_rank = [&](int r) { int c = fn(r); if (c && !rc) { rc = c; err = mrt_last_error(); } };
        inline_rank(0);
        for (int r = par + 1; r < ranks; r++) inline_rank(r);
        for (int r = 1; r <= par; r++) {
            Worker *k = w[(size_t)r - 1].get();
            std::unique_lock<std::mutex> l(k->m); k->cv.wait(l, [&] { return k->done; });
            if (k->rc && !rc) { rc = k->rc; err = k->err; }
        }
        if (rc) mrt::set_error(err);
        return rc;
    }
};

}  // namespace

enum { MRT_REDUCE_RCCL = 0, MRT_REDUCE_PEER_COPY = 1, MRT_REDUCE_COMPACT = 2 };

struct MRTGroup_ {
    std::vector<int> devices;
    std::vector<MRTContext> ctx;          // owned
    std::vector<ncclComm_t> comms;        // one per rank when the group reduces with RCCL
    int reduce = MRT_REDUCE_PEER_COPY;
    std::string reduce_note;
    RankPool pool;                        // one host thread per device (rank 0: the caller's)
};
struct MRTGroupRenderer_ {
    MRTGroup g = nullptr;
    std::vector<MRTScene> scenes;         // owned replicas, one per rank
    std::vector<MRTRenderer> r;           // owned
    std::vector<hipEvent_t> done;         // per rank: recorded on its stream before the gather
    int width = 0, height = 0;
    mrt::DevBuf<float4> gathered, staging;   // on the root device
    std::vector<std::unique_ptr<mrt::DevBuf<float4>>> packed;      // reduce mode 2: per rank >= 1, on its own device: the compact buffer of its tiles
};

extern "C" {

// -------------------------------------------------------------------------------------------------- group
int mrt_group_create(const int *device_ids, int32_t n, MRTGroup *out) {
    MRT_TRY
    REQUIRE(out, "mrt_group_create: out is NULL");
    *out = nullptr;
    REQUIRE(device_ids && n >= 1 && n <= 64, "mrt_group_create: need 1 <= n <= 64 device ids");
    std::unique_ptr<MRTGroup_> g(new MRTGroup_());
    struct Undo { MRTGroup_ *g; ~Undo() { if (g) { for (auto c : g->ctx) mrt_context_destroy(c); } } } undo{g.get()};
    for (int i = 0; i < n; i++) {
        MRTContext c = nullptr;
        int rc = mrt_context_create(device_ids[i], &c); if (rc) return rc;
        g->ctx.push_back(c); g->devices.push_back(device_ids[i]);
    }
    const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == (size_t)n;
    if (n == 1) g->reduce_note = "one device: nothing to reduce";
    else if (!distinct) g->reduce_note = "a device is named more than once: peer copies + add (RCCL refuses duplicate devices)";
    else if (!g_rccl.load()) g->reduce_note = "peer copies + add (" + g_rccl.why + ")";
    else {
        g->comms.assign((size_t)n, nullptr);
        MRT_RCCL(g_rccl.CommInitAll(g->comms.data(), n, g->devices.data()));
        g->reduce = MRT_REDUCE_RCCL; g->reduce_note = "ncclReduce(sum, float32) to rank 0";
    }
    g->pool.start(n);
    undo.g = nullptr;
    *out = g.release();
    return MRT_OK;
    MRT_CATCH
}
int mrt_group_destroy(MRTGroup g) {
    if (!g) return MRT_OK;
    for (size_t i = 0; i < g->comms.size(); i++) if (g->comms[i]) { (void)hipSetDevice(g->devices[i]); (void)g_rccl.CommDestroy(g->comms[i]); }
    for (auto c : g->ctx) mrt_context_destroy(c);
    delete g;
    return MRT_OK;
}
int mrt_group_size(MRTGroup g, int32_t *n) { REQUIRE(g && n, "mrt_group_size: bad argument"); *n = (int32_t)g->ctx.size(); return MRT_OK; }
int mrt_group_context(MRTGroup g, int32_t rank, MRTContext *ctx) {
    REQUIRE(g && ctx && rank >= 0 && (size_t)rank < g->ctx.size(), "mrt_group_context: bad argument");
    *ctx = g->ctx[(size_t)rank];
    return MRT_OK;
}
// how the group assembles an image: 0 = ncclReduce, 1 = peer copies + add, 2 = compact (owned tiles only); the text says why
int mrt_group_reduce_mode(MRTGroup g, int32_t *mode, char *note, size_t note_len) {
    REQUIRE(g && mode, "mrt_group_reduce_mode: bad argument");
    *mode = g->reduce;
    if (note && note_len) snprintf(note, note_len, "%s", g->reduce_note.c_str());
    return MRT_OK;
}
int mrt_group_set_reduce_mode(MRTGroup g, int32_t mode) {
    REQUIRE(g && (mode == MRT_REDUCE_RCCL || mode == MRT_REDUCE_PEER_COPY || mode == MRT_REDUCE_COMPACT), "mrt_group_set_reduce_mode: mode must be 0 (RCCL reduce), 1 (peer copies + add) or 2 (compact: owned tiles only)");
    if (mode == MRT_REDUCE_COMPACT) {          // the transport is what the group has: ncclSend / ncclRecv with communicators, peer copies without
        g->reduce = mode;
        g->reduce_note = std::string("compact: every rank ships the tiles it owns (1/n of the image), the root writes them in place; transport: ") + ((!g->comms.empty() && g_rccl.Send && g_rccl.Recv) ? "ncclSend / ncclRecv" : "peer copies");
        return MRT_OK;
    }
    if (mode == MRT_REDUCE_RCCL && g->comms.empty()) {
        // a one-device group has no communicator until it is asked for one (ncclReduce over one rank is a copy: it exercises the RCCL path on a one-GPU box)
        const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == g->devices.size();
        if (!distinct) { mrt::set_error("mrt_group_set_reduce_mode: this group has no RCCL communicators (" + g->reduce_note + ")"); return MRT_ERR_UNSUPPORTED; }
        if (!g_rccl.load()) { mrt::set_error("mrt_group_set_reduce_mode: " + g_rccl.why); return MRT_ERR_UNSUPPORTED; }
        g->comms.assign(g->devices.size(), nullptr);
        ncclResult_t e = g_rccl.CommInitAll(g->comms.data(), (int)g->devices.size(), g->devices.data());
        if (e != ncclSuccess) { g->comms.clear(); return rccl_fail(e, "ncclCommInitAll"); }
        g->reduce_note = "ncclReduce(sum, float32) to rank 0";
    }
    g->reduce = mode;
    return MRT_OK;
}

// -------------------------------------------------------------------------------------------------- sharded renderer
int mrt_group_renderer_destroy(MRTGroupRenderer gr) {
    if (!gr) return MRT_OK;
    for (size_t i = 0; i < gr->r.size(); i++) if (gr->r[i]) mrt_renderer_destroy(gr->r[i]);
    for (size_t i = 0; i < gr->done.size(); i++) if (gr->done[i]) { (void)hipSetDevice(gr->g->devices[i]); (void)hipEventDestroy(gr->done[i]); }
    for (size_t i = 0; i < gr->scenes.size(); i++) if (gr->scenes[i]) mrt_scene_destroy(gr->scenes[i]);
    if (!gr->g->devices.empty()) (void)hipSetDevice(gr->g->devices[0]);
    delete gr;
    return MRT_OK;
}
int mrt_group_renderer_create(MRTGroup g, MRTScene scene, int32_t width, int32_t height, uint32_t seed, int32_t max_bounces, MRTGroupRenderer *out) {
    MRT_TRY
    REQUIRE(g && scene && out, "mrt_group_renderer_create: bad argument");
    *out = nullptr;
    std::unique_ptr<MRTGroupRenderer_> gr(new MRTGroupRenderer_());
    struct Undo { MRTGroupRenderer_ *p; ~Undo() { if (p) mrt_group_renderer_destroy(p); } } undo{gr.get()};
    gr->g = g; gr->width = width; gr->height = height;
    const int n = (int)g->ctx.size();
    MRTGroupRenderer_ *raw = gr.release();        // from here on the Undo guard owns it
    raw->scenes.assign((size_t)n, nullptr); raw->r.assign((size_t)n, nullptr); raw->done.assign((size_t)n, nullptr);
    // every rank on its own host thread: eight BVH builds (and the eight uploads in front of them) run side by side instead of one after the other
    int rc_all = g->pool.run(n, [&](int rank) -> int {
        // the scene is replicated: the caller's meshes, lights and build options, committed (BVH built) on this rank's device
        MRTScene s = nullptr;
        int rc = mrt_scene_create(g->ctx[(size_t)rank], &s); if (rc) return rc;
        raw->scenes[(size_t)rank] = s;
        s->meshes = scene->meshes; s->lights = scene->lights; s->opt = scene->opt;
        rc = mrt_scene_commit(s); if (rc) return rc;
        MRTRenderer r = nullptr;
        rc = mrt_renderer_create(g->ctx[(size_t)rank], s, width, height, seed, max_bounces, &r); if (rc) return rc;
        raw->r[(size_t)rank] = r;
        rc = mrt_renderer_set_shard(r, rank, n); if (rc) return rc;
        // a shard's launches are 1/n of a frame: carry proportionally more frames per pass so that they stay large
        rc = mrt_renderer_set_option(r, "frame_batch", (double)std::min(mrt::MAX_FRAME_BATCH, mrt::DEFAULT_FRAME_BATCH * n)); if (rc) return rc;
        MRT_HIP(hipSetDevice(g->devices[(size_t)rank]));
        hipEvent_t e = nullptr; MRT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        raw->done[(size_t)rank] = e;
        return MRT_OK;
    });
    if (rc_all) return rc_all;
    MRT_HIP(hipSetDevice(g->devices[0]));
    MRT_HIP(raw->gathered.alloc((size_t)width * height));
    undo.p = nullptr;
    *out = raw;
    return MRT_OK;
    MRT_CATCH
}
int mrt_group_renderer_rank(MRTGroupRenderer gr, int32_t rank, MRTRenderer *r) {
    REQUIRE(gr && r && rank >= 0 && (size_t)rank < gr->r.size(), "mrt_group_renderer_rank: bad argument");
    *r = gr->r[(size_t)rank];
    return MRT_OK;
}
int mrt_group_set_option(MRTGroupRenderer gr, const char *key, double value) {
    REQUIRE(gr && key, "mrt_group_set_option: bad argument");
    for (auto r : gr->r) { int rc = mrt_renderer_set_option(r, key, value); if (rc) return rc; }
    return MRT_OK;
}
int mrt_group_set_camera(MRTGroupRenderer gr, const MRTCamera *camera) {
    REQUIRE(gr && camera, "mrt_group_set_camera: bad argument");
    for (auto r : gr->r) { int rc = mrt_renderer_set_camera(r, camera); if (rc) return rc; }
    return MRT_OK;
}
// draw(in:) n_frames times on every device of the group: enqueues and returns (the devices run concurrently)
int mrt_group_render(MRTGroupRenderer gr, int32_t n_frames) {
    REQUIRE(gr, "mrt_group_render: renderer is NULL");
    // enqueueing a draw is 0.2-0.4 ms of host time per device (40 launches, memsets, events for 20 frames): from one thread the eighth device would start ~3 ms after
    // the first — longer than a rank of eight works on 20 frames.  Every device's draw is enqueued from its own host thread.
    return gr->g->pool.run((int)gr->r.size(), [&](int rank) -> int { return mrt_renderer_render(gr->r[(size_t)rank], n_frames); });
}
int mrt_group_wait(MRTGroupRenderer gr) {
    REQUIRE(gr, "mrt_group_wait: renderer is NULL");
    for (auto r : gr->r) { int rc = mrt_renderer_wait(r); if (rc) return rc; }
    return MRT_OK;
}
// frames every device has finished (the minimum over the group); never blocks
int mrt_group_frames_completed(MRTGroupRenderer gr, uint64_t *frames) {
    REQUIRE(gr && frames, "mrt_group_frames_completed: bad argument");
    uint64_t m = ~0ull;
    for (auto r : gr->r) { uint64_t f = 0; int rc = mrt_renderer_frames_completed(r, &f); if (rc) return rc; m = std::min(m, f); }
    *frames = gr->r.empty() ? 0 : m;
    return MRT_OK;
}
// ONE reduce(sum) of the RGBA32F accumulation buffers into the root device (rank 0), enqueued behind the frames already submitted;
// rgba (host, width*height*16 bytes, row 0 = bottom of the image) may be NULL: the image then stays on the root device
// (mrt_group_gathered_device_ptr).  Blocks until the image is assembled.
int mrt_group_gather(MRTGroupRenderer gr, float *rgba, size_t nbytes) {
    MRT_TRY
    REQUIRE(gr, "mrt_group_gather: renderer is NULL");
    const size_t npix = (size_t)gr->width * gr->height, bytes = npix * sizeof(float4);
    REQUIRE(rgba == nullptr || nbytes == bytes, "mrt_group_gather: nbytes must be width*height*16");
    MRTGroup g = gr->g;
    const int n = (int)gr->r.size();
    hipStream_t s0 = g->ctx[0]->stream;
    auto src = [&](int rank) { mrt::Renderer &R = gr->r[(size_t)rank]->r; return R.accum[R.cur].p; };
    if (g->reduce == MRT_REDUCE_COMPACT) {
        // every rank >= 1 packs the tiles it owns on its own device and stream (behind its frames); the root takes its own buffer whole (its other pixels are 0 and are
        // overwritten), receives the n - 1 compact buffers side by side in its staging area and writes each shard's tiles in place
        const bool rccl = !g->comms.empty() && g_rccl.Send && g_rccl.Recv;
        if (gr->packed.size() != (size_t)n) { gr->packed.clear(); for (int k = 0; k < n; k++) gr->packed.emplace_back(new mrt::DevBuf<float4>()); }
        std::vector<size_t> off((size_t)n + 1, 0);
        for (int rank = 1; rank < n; rank++) off[(size_t)rank + 1] = off[(size_t)rank] + (size_t)mrt::shard_tiles(gr->width, gr->height, rank, n) * 64;
        for (int rank = 1; rank < n; rank++) {
            const size_t cnt = off[(size_t)rank + 1] - off[(size_t)rank];
            MRT_HIP(hipSetDevice(g->devices[(size_t)rank]));
            MRT_HIP(gr->packed[(size_t)rank]->alloc(std::max<size_t>(cnt, 1)));
            int rc = gr->r[(size_t)rank]->r.pack_owned_tiles(gr->packed[(size_t)rank]->p, cnt * sizeof(float4)); if (rc) return rc;
            MRT_HIP(hipEventRecord(gr->done[(size_t)rank], g->ctx[(size_t)rank]->stream));
        }
        MRT_HIP(hipSetDevice(g->devices[0]));
        MRT_HIP(hipMemcpyAsync(gr->gathered.p, src(0), bytes, hipMemcpyDeviceToDevice, s0));
        if (n > 1 && gr->staging.n < std::max<size_t>(off[(size_t)n], 1)) MRT_HIP(gr->staging.alloc(std::max<size_t>(off[(size_t)n], 1)));
        if (rccl && n > 1) {
            MRT_RCCL(g_rccl.GroupStart());
            for (int rank = 1; rank < n; rank++) {
                const size_t cnt = off[(size_t)rank + 1] - off[(size_t)rank];
                if (!cnt) continue;
                MRT_HIP(hipSetDevice(g->devices[(size_t)rank]));
                ncclResult_t e = g_rccl.Send(gr->packed[(size_t)rank]->p, cnt * 4, ncclFloat, 0, g->comms[(size_t)rank], g->ctx[(size_t)rank]->stream);
                if (e != ncclSuccess) { (void)g_rccl.GroupEnd(); return rccl_fail(e, "ncclSend"); }
                MRT_HIP(hipSetDevice(g->devices[0]));
                e = g_rccl.Recv(gr->staging.p + off[(size_t)rank], cnt * 4, ncclFloat, rank, g->comms[0], s0);
                if (e != ncclSuccess) { (void)g_rccl.GroupEnd(); return rccl_fail(e, "ncclRecv"); }
            }
            MRT_RCCL(g_rccl.GroupEnd());
            for (int rank = 1; rank < n; rank++) { MRT_HIP(hipSetDevice(g->devices[(size_t)rank])); MRT_HIP(hipStreamSynchronize(g->ctx[(size_t)rank]->stream)); }
            MRT_HIP(hipSetDevice(g->devices[0]));
        } else {
            for (int rank = 1; rank < n; rank++) {
                const size_t cnt = off[(size_t)rank + 1] - off[(size_t)rank];
                MRT_HIP(hipStreamWaitEvent(s0, gr->done[(size_t)rank], 0));
                if (cnt) MRT_HIP(hipMemcpyPeerAsync(gr->staging.p + off[(size_t)rank], g->devices[0], gr->packed[(size_t)rank]->p, g->devices[(size_t)rank], cnt * sizeof(float4), s0));
            }
        }
        for (int rank = 1; rank < n; rank++) {
            const size_t cnt = off[(size_t)rank + 1] - off[(size_t)rank];
            int rc = mrt::unpack_tiles_into(gr->gathered.p, gr->width, gr->height, gr->staging.p + off[(size_t)rank], cnt * sizeof(float4), rank, n, s0); if (rc) return rc;
        }
    } else if (g->reduce == MRT_REDUCE_RCCL && !g->comms.empty()) {
        MRT_RCCL(g_rccl.GroupStart());
        for (int rank = 0; rank < n; rank++) {
            MRT_HIP(hipSetDevice(g->devices[(size_t)rank]));
            ncclResult_t e = g_rccl.Reduce(src(rank), rank == 0 ? (void *)gr->gathered.p : nullptr, npix * 4, ncclFloat, ncclSum, 0, g->comms[(size_t)rank], g->ctx[(size_t)rank]->stream);
            if (e != ncclSuccess) { (void)g_rccl.GroupEnd(); return rccl_fail(e, "ncclReduce"); }
        }
        MRT_RCCL(g_rccl.GroupEnd());
        for (int rank = 1; rank < n; rank++) { MRT_HIP(hipSetDevice(g->devices[(size_t)rank])); MRT_HIP(hipStreamSynchronize(g->ctx[(size_t)rank]->stream)); }
        MRT_HIP(hipSetDevice(g->devices[0]));
    } else {
        // peer copies: the root waits for every rank's frames, pulls that rank's buffer and adds it
        for (int rank = 1; rank < n; rank++) { MRT_HIP(hipSetDevice(g->devices[(size_t)rank])); MRT_HIP(hipEventRecord(gr->done[(size_t)rank], g->ctx[(size_t)rank]->stream)); }
        MRT_HIP(hipSetDevice(g->devices[0]));
        MRT_HIP(hipMemcpyAsync(gr->gathered.p, src(0), bytes, hipMemcpyDeviceToDevice, s0));
        if (n > 1 && gr->staging.n < npix) MRT_HIP(gr->staging.alloc(npix));
        for (int rank = 1; rank < n; rank++) {
            MRT_HIP(hipStreamWaitEvent(s0, gr->done[(size_t)rank], 0));
            MRT_HIP(hipMemcpyPeerAsync(gr->staging.p, g->devices[0], src(rank), g->devices[(size_t)rank], bytes, s0));
            hipLaunchKernelGGL(k_add4, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s0, gr->gathered.p, gr->staging.p, npix);
        }
    }
    MRT_HIP(hipStreamSynchronize(s0));
    MRT_HIP(hipGetLastError());
    if (rgba) MRT_HIP(hipMemcpy(rgba, gr->gathered.p, bytes, hipMemcpyDeviceToHost));
    return MRT_OK;
    MRT_CATCH
}
int mrt_group_gathered_device_ptr(MRTGroupRenderer gr, void **device_ptr) {
    REQUIRE(gr && device_ptr, "mrt_group_gathered_device_ptr: bad argument");
    *device_ptr = gr->gathered.p;
    return MRT_OK;
}
// ray counters and frames of the whole group (sums over the ranks; `frames` and the timing fields are rank 0's)
int mrt_group_stats(MRTGroupRenderer gr, MRTRenderStats *out) {
    REQUIRE(gr && out, "mrt_group_stats: bad argument");
    memset(out, 0, sizeof *out);
    for (size_t i = 0; i < gr->r.size(); i++) {
        MRTRenderStats s; int rc = mrt_renderer_stats(gr->r[i], &s); if (rc) return rc;
        out->closest_rays += s.closest_rays; out->shadow_rays += s.shadow_rays; out->primary_rays += s.primary_rays; out->bytes_alg += s.bytes_alg;
        if (i == 0) { out->frames = s.frames; out->ms_gpu_last = s.ms_gpu_last; out->ms_extend_last = s.ms_extend_last; out->extend_launches_last = s.extend_launches_last; }
        else out->ms_gpu_last = std::max(out->ms_gpu_last, s.ms_gpu_last);
    }
    return MRT_OK;
}

}  // extern "C"
