#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: Mrays/s (closest-hit + shadow rays) and ms/frame,
DragonScene 1920x1080 spp=1 (configs[1]).

A "step" is one frame (one pass of the hot path over all pixels at 1 spp): primary rays, then per bounce
shade / trace (closest hit for the bounce rays + any hit for the shadow rays), then accumulate —
`Renderer.draw(in:)` of the reference (Renderer.swift:284-351).  Inputs (scene, BVH, seeds) are resident in HBM
before the timed region.  The renderer carries the K steps in passes of `frame_batch` frames on
`frames_in_flight` HIP streams (the reference keeps 3 frames in flight, Renderer.swift:33); every frame is
rendered in full and the running average is applied in frame order.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: one process per GPU, scene + BVH replicated.  Default `--shard tile` (BASELINE.json north_star): the image is
sharded by 8x8 screen tile (tile_id % N == rank, total work fixed -> "strong" scaling); `--shard sample`: every rank
renders full frames of a disjoint Halton index range (per-GPU work fixed -> "weak").  Either way the K frames are
accumulated locally and ONE RCCL reduce of the RGBA32F radiance buffer assembles the image on rank 0 inside the timed
region.

Prints ONE JSON line on rank 0.  Besides the driver's keys:
  roofline        dominant kernel (bounce + shadow traversal) ALONE on the chip at the default pass size: algorithmic bytes of one
                  8-frame launch / its average launch duration (the kernel's own start/stop events on its launching stream, one
                  stream, nothing else in flight — measured in this run right after the timed region; profiles/ holds rocprofv3's
                  view of the same regime) against HBM 8 TB/s.  `under_overlap` is the same kernel's launch duration inside the
                  timed region, where up to frames_in_flight passes share the chip (a regime, not a kernel time); `frame` is
                  SURVEY §8(d)'s frame-level figure bytes_alg / t_frame.
  latency         SURVEY §8(d)'s ms/frame: device time of all kernels of ONE frame (frames_in_flight = 1, frame_batch = 1,
                  median of 24), the serialised per-kernel times, and the reference's own mode (3 frames in flight).
  calibration     v_fma_f32 issue rate and divergent-gather rate CALIBRATED on this chip in this run (mrt_debug_calibrate).
  cpu_baseline    the CPU oracle on the host cores, one full frame (reported, not the target).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # one hardware queue per frame in flight (before HIP initialises)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PASS_FRAMES = 8                # the library's default frame_batch: the serialised-pass leg and the committed counters use launches of this many frames
BYTES_PER_CLOSEST_RAY = 96     # ray 2x32 B + hit 2x16 B
BYTES_PER_SHADOW_RAY = 72


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "dragon_irregular", "dragon_hostile", "cornell", "dragon4", "garden"])
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--shard", default="tile", choices=["tile", "sample"], help="N > 1: 8x8 screen-tile sharding (north_star, strong scaling, default) or sample-index sharding (weak scaling)")
    ap.add_argument("--builder", type=int, default=None)
    ap.add_argument("--opt", action="append", default=[], help="renderer option key=value (repeatable)")
    ap.add_argument("--sopt", action="append", default=[], help="scene (BVH build) option key=value (repeatable)")
    ap.add_argument("--frames-in-flight", type=int, default=None, help="passes in flight on separate HIP streams (library default 3, each carrying frame_batch frames: 8 at 1080p and above, up to 32 for smaller images; the reference keeps 3 frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="skip the extra max_bounces=1 (primary + shadow only) measurement")
    ap.add_argument("--no-latency", action="store_true", help="skip the serialised per-frame latency leg and the on-chip calibration")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal on one GPU box (ranks share GPUs, reduce on host)")
    ap.add_argument("--group-devices", default=None, help="--gpus N without torch.distributed.run: device ids of the C-ABI device group (default 0..N-1; '0,0' rehearses N = 2 on a one-GPU box)")
    ap.add_argument("--assemble", default="reduce", choices=["reduce", "compact"], help="N > 1, tile sharding: how the image is assembled on rank 0 — the ONE reduce(sum) of the whole RGBA32F buffers (north_star, default) or the compact form: every rank ships only the tiles it owns")
    ap.add_argument("--steady-steps", type=int, default=240, help="N > 1: after the timed region, one more leg of this many steps (not part of value / ms_per_step) reported as \"steady\": the regime a long run reaches, beside the launch-bound figure of a short --steps (0 = skip)")
    ap.add_argument("--png", default=None, help="write the tonemapped image here (rank 0)")
    ap.add_argument("--dump-accum", default=None, help="write the assembled RGBA32F accumulation buffer (warm-up + timed frames) here as .npy (rank 0)")
    return ap.parse_args()


def csrc_hash():
    """sha256 over the library's sources (metal-raytracing_amd/csrc/*.{hip,h,cpp}, Makefile): tools/summarize_profiles.py stores it with the counters
    it summarises, and the counter-derived fields of the bench line are emitted only when the running tree has the same hash."""
    import glob, hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "metal-raytracing_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def committed_profile():
    """Per-dispatch HBM bytes of the dominant kernel and VALU instructions per frame from the rocprofv3 PMC passes kept under
    profiles/ (tools/collect_profiles.sh; FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md §HBM prescribes for gfx950; gather widths uncalibrated, so an upper estimate).  {} if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
        d["_source"] = os.path.relpath(files[-1], ROOT)
        return d
    except Exception:
        return {}


def cpu_baseline(mrt, scene, w, h, bounces, threads, two_level=False):
    """The oracle (CPU restatement, kind 'port') on the GPU box's host cores: one full frame of the
    same workload, same seeds.  Reported, never the thing shipped.  SURVEY §8(d): built here, on this box, with -O3 -march=native
    (oracle/Makefile target `native`; no fast-math, no contraction: the same bits as the portable build the tests use) and run on
    all host cores.  It is the stated non-target baseline; nothing rides on the ratio."""
    import oracle as O                             # (built by prebuild_oracle() before anything touched the GPU: -O3 -march=native, recompiled on this box, or the portable build if that failed)
    threads = threads or (os.cpu_count() or 1)     # all host cores of the box
    # a two-level scene (--sopt instancing=1) is checked against the oracle's two-level restatement: triangle tests in object space round differently from the flattened scene's
    osc = O.OracleScene(mrt.flatten_scene(scene, share=True), scene.lights, instancing=True) if two_level else O.OracleScene(mrt.flatten_scene(scene), scene.lights)
    r = O.OracleRenderer(osc, w, h, seed=1, max_bounces=bounces, camera=scene.camera)
    t0 = time.perf_counter()
    r.render(1, threads=threads)
    dt = time.perf_counter() - t0
    closest, shadow = r.counters()
    img = r.accumulation()
    return {"value": (closest + shadow) / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "build": O.ORACLE_BUILD,
            "sample": f"1 full frame of the same workload ({w}x{h} spp=1, {bounces} bounces, {closest + shadow} rays) in {dt:.2f} s",
            "ms_per_frame": dt * 1e3}, img


def prebuild_oracle():
    """The cpu_baseline leg's oracle, compiled BEFORE the GPU legs (no child process of a parent that holds the GPU, and a failing compiler cannot cost the GPU result):
    -O3 -march=native recompiled on this box (`make -B native`: a stale binary from another host could be built for another CPU); if that fails, the portable build
    the tests use.  Returns a note for the JSON line, or None."""
    import importlib
    os.environ["MRT_ORACLE_NATIVE"] = "1"
    try:
        import oracle as O
        O.build_oracle(force=True)
        return None
    except Exception as e:
        os.environ["MRT_ORACLE_NATIVE"] = "0"
        note = f"native oracle build failed ({type(e).__name__}: {e}); "
        try:
            import oracle as O
            importlib.reload(O)
            O.build_oracle()
            return note + "cpu_baseline uses the portable -O2 -march=x86-64-v3 build"
        except Exception as e2:
            return note + f"portable build failed too ({type(e2).__name__}: {e2}): no cpu_baseline"


def latency_leg(mrt, r, scene, w, h, bounces, opts, frames=24):
    """SURVEY §8(d) ms/frame: hipEvent time of all kernels of ONE frameIndex.  A second renderer on the same context and scene
    options, one frame per draw call, nothing else in flight; then the reference's own regime (3 frames in flight, one frame
    per pass, Renderer.swift:33)."""
    q = mrt.Renderer((w, h), scene, ctx=r.ctx, seed=1, max_bounces=bounces, scene_options=opts)
    q.set_option("frames_in_flight", 1); q.set_option("frame_batch", 1)
    def one_frame_alone(groups):
        q.set_option("tile_groups", groups)
        q.draw(3, wait=True)
        per_frame, per_kernel = [], {}
        for _ in range(frames):
            q.draw(1, wait=True)
            per_frame.append(q.stats.ms_gpu_last)
            for k, (ms, n) in q.kernel_times.items():
                if n:
                    per_kernel.setdefault(k, []).append(ms / n)
        return per_frame, per_kernel, int(q.get_option("groups_used"))
    # the frame as ONE pass on one stream (tile_groups = 1): seven dependent launches, each alone on the chip — the per-kernel times that are kernel times
    whole, per_kernel, _ = one_frame_alone(1)
    # the library's default for a lone frame: the pass as tile groups on several lanes (Renderer::tile_groups), one group's shade beside another's traversal
    per_frame, _, groups = one_frame_alone(0)
    out = {"ms_per_frame": round(statistics.median(per_frame), 4), "min": round(min(per_frame), 4), "max": round(max(per_frame), 4), "frames": frames, "tile_groups_used": groups,
           "mode": "frames_in_flight=1 frame_batch=1: begin-to-end device time of one frame, nothing else on the GPU (the pass runs as tile_groups_used groups of tiles on as many HIP streams)",
           "ms_per_frame_as_one_pass": round(statistics.median(whole), 4),
           "kernel_ms_serialised": {k: round(statistics.median(v), 4) for k, v in per_kernel.items()}}
    # the default pass size alone on the GPU: one stream, passes of PASS_FRAMES frames (the regime in which the dominant kernel's standalone duration is
    # consistent with ms_per_step; profiles/r03_kernel_stats_serial_pass.csv is rocprofv3's view of the same regime)
    q.set_option("frame_batch", PASS_FRAMES); q.set_option("tile_groups", 1); q.draw(2 * PASS_FRAMES, wait=True)          # (whole passes, one after the other: a kernel alone on the chip)
    tot = {}
    for _ in range(5):
        q.draw(2 * PASS_FRAMES, wait=True)   # two passes per call: the grid policy of a long call (half the wave slots per traversal launch), as in the default run and in the rocprofv3 trace
        for k, (ms, n) in q.kernel_times.items():
            if n:
                t = tot.setdefault(k, [0.0, 0]); t[0] += ms; t[1] += n
    # the AVERAGE over all launches of a class (the three traversal launches of a pass carry different ray counts: rocprofv3 --stats averages the same way)
    out["kernel_ms_serialised_pass"] = {k: round(ms / n, 4) for k, (ms, n) in tot.items()}; out["launches_serialised_pass"] = {k: n for k, (ms, n) in tot.items()}; out["frames_per_serialised_pass"] = PASS_FRAMES
    q.set_option("frame_batch", 1); q.set_option("tile_groups", 0); q.draw(2, wait=True)
    # the same single frame as ONE launch (k_megakernel: whole paths per lane, no queues; lowest latency, lower throughput)
    if q.device_scene.stats.wide_layout and not (opts or {}).get("instancing"):      # (the one-launch mode renders flattened scenes on the 8-wide layout; it refuses others)
        q.set_option("megakernel", 1); q.draw(3, wait=True)
        mk = []
        for _ in range(frames):
            q.draw(1, wait=True); mk.append(q.stats.ms_gpu_last)
        out["megakernel_ms_per_frame"] = round(statistics.median(mk), 4)
        q.set_option("megakernel", 0)
    q.set_option("frames_in_flight", 3)
    q.draw(6, wait=True)
    t0 = time.perf_counter(); q.draw(30, wait=True); dt = time.perf_counter() - t0
    out["reference_like_3_in_flight_ms_per_frame"] = round(dt * 1e3 / 30, 4)
    # last (the option cannot be taken back): the serialised passes again with the pulling launches on ALL the wave slots — a long call runs them on half, so that the other passes'
    # launches find free slots; alone on the chip the kernel is faster on all of them.  roofline.whole_chip
    q.set_option("frames_in_flight", 1); q.set_option("frame_batch", PASS_FRAMES); q.set_option("tile_groups", 1); q.set_option("wave_slots", q.get_option("wave_slots"))
    q.draw(2 * PASS_FRAMES, wait=True)
    tot = {}
    for _ in range(5):
        q.draw(2 * PASS_FRAMES, wait=True)
        for k, (ms, n) in q.kernel_times.items():
            if n:
                t = tot.setdefault(k, [0.0, 0]); t[0] += ms; t[1] += n
    out["kernel_ms_serialised_pass_whole_chip"] = {k: round(ms / n, 4) for k, (ms, n) in tot.items()}; out["wave_slots"] = int(q.get_option("wave_slots"))
    q.close()
    return out


def main_group(a):
    """`python bench.py --gpus N` WITHOUT torch.distributed.run: one process drives the N GPUs through the C ABI's device group
    (include/mrt_abi.h mrt_group_*: replicated scene, tile_id % N shards, ONE ncclReduce(sum) per output image, RCCL opened by the library).
    The driver's launch (torch.distributed.run, one rank per GPU) takes the other path; both report the same metric."""
    import metal_raytracing_amd as mrt
    w, h = a.width, a.height
    scene = mrt.SCENES[a.scene]((w, h))
    opts = {} if a.builder is None else {"builder": a.builder}
    for kv in a.sopt:
        k, v = kv.split("="); opts[k] = float(v)
    devices = [int(x) for x in a.group_devices.split(",")] if a.group_devices else list(range(a.gpus))
    if len(devices) != a.gpus:
        sys.exit("--group-devices must name --gpus devices")
    g = mrt.GroupRenderer((w, h), scene, devices, seed=1, max_bounces=a.bounces, scene_options=opts)
    for kv in a.opt:
        k, v = kv.split("="); g.set_option(k, float(v))
    if a.frames_in_flight is not None:
        g.set_option("frames_in_flight", a.frames_in_flight)
    if a.assemble == "compact":
        g.set_reduce_mode(2)
    g.draw(a.warmup); g.gather(to_host=False)              # warm-up: W untimed steps and one reduce (communicator set-up is not timed)
    first = g.stats
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.draw(a.steps)                                         # enqueued on every device, returns at once
    g.wait()                                                # every device's frames done ...
    t_render = time.perf_counter() - t0
    g.gather(to_host=False)                                 # ... then the ONE reduce of the RGBA32F buffer into device 0
    dt = time.perf_counter() - t0
    per_rank = [g.rank_stats(k) for k in range(a.gpus)]     # what the first real SCALE record is diagnosed with: which rank the draw waited for, and what the reduce cost
    st = g.stats
    closest, shadow, primary = st.closest_rays - first.closest_rays, st.shadow_rays - first.shadow_rays, st.primary_rays - first.primary_rays
    mode, note = g.reduce_mode
    out = {"metric": "Mrays/sec (primary+shadow) and ms/frame, DragonScene 1920x1080 spp=1" if (a.scene, w, h) == ("dragon", 1920, 1080) else f"Mrays/sec (closest+shadow), {a.scene} {w}x{h} spp=1",
           "value": round((closest + shadow) / dt / 1e6, 3), "unit": "Mrays/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(dt * 1e3 / a.steps, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{a.scene} scene {w}x{h} spp=1, {a.bounces} bounces, closest-hit + shadow rays counted on device", "scene_sources": scene.describe(),
                      "launch": "one process, C-ABI device group (mrt_group_*)", "shard": "tile", "reduce": note, "reduce_mode": mode,
                      "frame_batch": int(g.rank_option(0, "frame_batch")), "frames_in_flight": int(g.rank_option(0, "frames_in_flight")),
                      "rays_per_frame": {"closest": closest / a.steps, "shadow": shadow / a.steps, "primary": primary / a.steps}},
           "per_rank": {"ms_gpu_timed_draw": [round(x.ms_gpu_last, 4) for x in per_rank], "rays_timed_plus_warmup": [int(x.closest_rays + x.shadow_rays) for x in per_rank],
                        "render_wall_ms": round(t_render * 1e3, 4), "gather_wall_ms": round((dt - t_render) * 1e3, 4),
                        "note": "ms_gpu_timed_draw: begin-to-end device time of the timed draw on each device (its own events); the draw's wall time is the slowest rank's plus the enqueue; gather_wall_ms is the one reduce(sum) of the image"},
           "roofline": None, "cpu_baseline": None}
    if a.dump_accum:
        np.save(a.dump_accum, g.gather())
    if a.steady_steps > 0 and a.gpus > 1:          # one more leg, not part of value: the regime a long run reaches
        before = g.stats
        torch.cuda.synchronize(); t1 = time.perf_counter()
        g.draw(a.steady_steps); g.wait(); t_r = time.perf_counter() - t1
        g.gather(to_host=False); dts = time.perf_counter() - t1
        s2 = g.stats
        out["steady"] = {"steps": a.steady_steps, "value": round((s2.closest_rays - before.closest_rays + s2.shadow_rays - before.shadow_rays) / dts / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dts * 1e3 / a.steady_steps, 4),
                         "per_rank": {"ms_gpu_timed_draw": [round(g.rank_stats(k).ms_gpu_last, 4) for k in range(a.gpus)], "render_wall_ms": round(t_r * 1e3, 4), "gather_wall_ms": round((dts - t_r) * 1e3, 4)},
                         "note": "a second leg after the timed region, same group, its own assemble: not part of value"}
    print(json.dumps(out), flush=True)
    g.close()


def launch_plan(gpus_arg, steps, warmup, shard, env, ndev, pixels=1920 * 1080):
    """What one process of a bench run does, from its arguments and environment alone (no GPU touched; tests/test_bench_contract.py checks it against DESIGN.md §7):
    torch.distributed.run sets WORLD_SIZE / RANK / LOCAL_RANK and those win over --gpus; without them --gpus N > 1 means ONE process driving N devices through
    the C ABI's device group.  Tile sharding: the rank's frame_batch and the passes its warm-up and timed draws split into."""
    from metal_raytracing_amd.distributed import shard_frame_batch, pass_sizes, auto_frame_batch
    world = int(env.get("WORLD_SIZE", "1")); rank = int(env.get("RANK", "0")); local_rank = int(env.get("LOCAL_RANK", "0"))
    plan = {"world": world, "rank": rank, "mode": "ranks" if world > 1 else ("group" if gpus_arg > 1 else "single"), "gpus": world if world > 1 else gpus_arg,
            "device": local_rank % max(1, ndev)}                                  # more ranks than GPUs: a gloo rehearsal on one box
    n = plan["gpus"]
    fb = shard_frame_batch(n, warmup + steps) if (shard == "tile" and n > 1) else auto_frame_batch(pixels)      # one device (or whole frames per rank): the library's default, by image size
    if plan["mode"] == "group":
        fb = shard_frame_batch(n)                                                 # mrt_group_renderer_create knows no run length; Renderer::render caps every draw itself
    plan.update(frame_batch=fb, warmup_passes=pass_sizes(warmup, fb) if warmup else [], timed_passes=pass_sizes(steps, fb))
    return plan


def main():
    a = parse()
    plan = launch_plan(a.gpus, a.steps, a.warmup, a.shard, os.environ, torch.cuda.device_count(), pixels=a.width * a.height)
    for kv in a.opt:                                  # an explicit --opt frame_batch=N is what the passes are made of
        if kv.startswith("frame_batch=") and float(kv.split("=")[1]) > 0 and plan["mode"] == "single":
            from metal_raytracing_amd.distributed import pass_sizes
            fb = int(float(kv.split("=")[1])); plan.update(frame_batch=fb, warmup_passes=pass_sizes(a.warmup, fb) if a.warmup else [], timed_passes=pass_sizes(a.steps, fb))
    world, rank, local_rank = plan["world"], plan["rank"], plan["device"]
    a.gpus = plan["gpus"]
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    elif plan["mode"] == "group":
        return main_group(a)             # one process, N devices: the C ABI's device group (mrt_group_*), no torch.distributed

    oracle_note = prebuild_oracle() if (world == 1 and not a.no_cpu_baseline) else None
    import metal_raytracing_amd as mrt
    from metal_raytracing_amd.distributed import ShardedRenderer
    w, h = a.width, a.height
    scene = mrt.SCENES[a.scene]((w, h))
    opts = {} if a.builder is None else {"builder": a.builder}
    for kv in a.sopt:
        k, v = kv.split("="); opts[k] = float(v)
    sr = ShardedRenderer((w, h), scene, rank, world, mode=a.shard, device=local_rank, frames_total=a.warmup + a.steps,
                         backend=a.dist_backend if world > 1 else None, seed=1, max_bounces=a.bounces, scene_options=opts)
    r = sr.renderer
    sst = r.device_scene.stats
    for kv in a.opt:
        k, v = kv.split("="); r.set_option(k, float(v))
    if a.frames_in_flight is not None:
        r.set_option("frames_in_flight", a.frames_in_flight)

    def sync():
        r.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- warmup (W untimed steps, and one reduce so that the collective's set-up is not timed either)
    sr.draw(a.warmup); r.wait()
    if world > 1:
        sr.gather()
    r.reset_stats()
    sync()
    compact = a.assemble == "compact" and a.shard == "tile"

    def timed_leg(steps):
        """`steps` steps bracketed by barrier + synchronize on both sides (+ the one assemble of the output image for N > 1); the MAX over ranks is the leg's time."""
        t0 = time.perf_counter()
        sr.draw(steps)                   # all steps are enqueued at once; the first 512 launches carry their own start/stop events
        r.wait()
        t_render = time.perf_counter() - t0
        if world > 1:
            sr.gather(compact=compact)
        t_gather = time.perf_counter() - t0 - t_render
        sync()
        dt = time.perf_counter() - t0
        st = r.stats
        kt = r.kernel_times
        per_rank = None
        if world > 1:                    # (outside the timed region) every rank's own device time, wall time to its last frame and time inside the assemble: what a SCALE record is diagnosed with
            mine = torch.tensor([st.ms_gpu_last, t_render * 1e3, t_gather * 1e3], dtype=torch.float64)
            if a.dist_backend == "nccl":
                mine = mine.cuda()
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = {"ms_gpu_timed_draw": [round(float(x[0]), 4) for x in allr], "render_wall_ms": [round(float(x[1]), 4) for x in allr], "gather_wall_ms": [round(float(x[2]), 4) for x in allr],
                        "assemble": "compact (owned tiles only: dist.gather of 1/N of the image per rank)" if compact else "reduce(sum) of the whole RGBA32F buffer",
                        "note": "per rank: device time of the timed draw (its own events), wall time until its last frame was done, wall time inside the one assemble of the image (a rank that arrives early waits there for the slowest)"}
        rays = torch.tensor([st.closest_rays, st.shadow_rays, st.primary_rays], dtype=torch.float64)
        tmax = torch.tensor([dt], dtype=torch.float64)
        if world > 1:
            if a.dist_backend == "nccl":
                rays = rays.cuda(); tmax = tmax.cuda()
            dist.all_reduce(rays); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            rays = rays.cpu(); tmax = tmax.cpu()
        return float(tmax[0]), st, kt, per_rank, rays

    # ---- timed region: exactly K steps (+ the one assemble of the output image for N > 1)
    dt, st, kt, per_rank, rays = timed_leg(a.steps)
    closest, shadow, primary = (float(x) for x in rays)
    # ---- N > 1: one more leg, NOT part of value / ms_per_step — the driver's short --steps is launch-bound on a rank of N (a few ms of work in ~20 dependent launches);
    # a long draw shows what the partition itself scales to
    steady = None
    dump_img = sr.buffer.cpu().numpy().copy() if (world > 1 and rank == 0 and a.dump_accum) else None      # the timed region's image: the steady leg goes on accumulating
    if world > 1 and a.steady_steps > 0:
        if a.shard == "tile":            # the pass size a run of that length takes (ShardedRenderer sized it for warm-up + timed steps); one untimed pass so that the queues are re-sized outside the leg
            from metal_raytracing_amd.distributed import shard_frame_batch
            fbs = shard_frame_batch(world, a.steady_steps)
            if fbs != int(r.get_option("frame_batch")):
                r.set_option("frame_batch", fbs); sr.draw(fbs); r.wait()
        r.reset_stats(); sync()
        dts, _, _, prs, rs = timed_leg(a.steady_steps)
        steady = {"steps": a.steady_steps, "frame_batch": int(r.get_option("frame_batch")), "value": round(float(rs[0] + rs[1]) / dts / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dts * 1e3 / a.steady_steps, 4), "per_rank": prs,
                  "note": "a second leg after the timed region, same renderer, same barriers, its own assemble: not part of value"}
    steps_total = a.steps * (world if a.shard == "sample" and world > 1 else 1)

    if rank == 0:
        value = (closest + shadow) / dt / 1e6
        fused = True                                                     # one traversal launch per bounce over [bounce rays | shadow rays]: the only pipeline the library has
        frame_batch = int(r.get_option("frame_batch"))
        if frame_batch > PASS_FRAMES:                                    # renderer.hip render(): passes larger than the default take at most a third of a draw
            frame_batch = min(frame_batch, max(PASS_FRAMES, (a.steps + 2) // 3))
        passes = (a.steps + frame_batch - 1) // frame_batch              # one pass of the pipeline = frame_batch frames
        # dominant kernel: the bounce + shadow traversal (k_trace_mixed_wide_persist): max_bounces launches per pass, each over
        # [bounce rays of that bounce (closest hit, 96 B) | shadow rays of that bounce (any hit, 72 B)]; the primary rays (96 B each)
        # belong to k_trace_primary.  This rank's rays (st.*), this rank's launches.
        trace_launches = a.bounces * passes if fused else a.bounces * a.steps
        traced_bytes = BYTES_PER_CLOSEST_RAY * (st.closest_rays - st.primary_rays) + (BYTES_PER_SHADOW_RAY * st.shadow_rays if fused else 0)
        bytes_per_launch = traced_bytes / max(1, trace_launches)
        rays_per_launch = (st.closest_rays - st.primary_rays + (st.shadow_rays if fused else 0)) / max(1, trace_launches)
        t_ms, t_n = kt["trace"]
        avg_ms = t_ms / max(1, t_n)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        frame_bytes = st.bytes_alg / max(1, st.frames)
        frame_gbs = frame_bytes * st.frames / dt / 1e9
        prof = committed_profile()
        default_opts = not a.opt and not a.sopt and a.builder is None and a.frames_in_flight is None
        single_dragon = world == 1 and (a.scene, w, h, a.bounces) == ("dragon", 1920, 1080, 3)
        # counters under profiles/ describe ONE configuration (dragon 1080p, default options, one GPU) of ONE source tree: anything else gets null
        prof_applies = bool(prof) and single_dragon and default_opts and prof.get("csrc_sha256") == csrc_hash()
        prof_note = None if prof_applies else ("no committed profile" if not prof else "committed counters not shown: " + ("they were collected on a different source tree (csrc hash differs; rerun tools/collect_profiles.sh)" if single_dragon and default_opts else "they describe dragon 1920x1080, 3 bounces, default options, one GPU — not this run"))
        two_level = bool(r.device_scene.stats.instances) and any(kv.startswith("instancing=1") for kv in a.sopt)
        if r.get_option("wide_bounce") == 0 or not sst.wide_layout:
            kernel_label = "k_trace_mixed (rope layout, bounce + shadow traversal)"
        else:
            pers = int(r.get_option("persistent"))
            pulls = pers == 1 or (pers == 2 and (2 * (r.stats.primary_rays / max(1, r.stats.frames)) * frame_batch >= r.get_option("wave_slots") * 1024 or (min(int(r.get_option("lanes_used")), passes) >= 5 and 2 * (r.stats.primary_rays / max(1, r.stats.frames)) * frame_batch >= r.get_option("wave_slots") * 256)))      # renderer.hip render(): the same rule
            kernel_label = ("k_trace_mixed_wide_persist" if pulls else "k_trace_mixed_wide_stream") + ("<two-level>" if two_level else "") + " (bounce + shadow traversal)"
            if two_level and r.get_option("tl_pairs") != 0 and a.bounces <= 3 and r.get_option("materials") == 0:
                kernel_label = "k_tl_top_flat / k_tl_top + k_tl_blas (two-level scene, binned: TLAS pass + BLAS pass over (ray, instance) pairs; two launches per bounce, timed together)"
        traffic_p = (prof.get("hbm_traffic_bytes_per_launch", {}).get("k_trace_mixed_wide_persist") or {}).get("bytes_corrected") if prof else None
        out = {
            "metric": "Mrays/sec (primary+shadow) and ms/frame, DragonScene 1920x1080 spp=1" if (a.scene, w, h) == ("dragon", 1920, 1080) else f"Mrays/sec (closest+shadow), {a.scene} {w}x{h} spp=1",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt * 1e3 / a.steps, 4), "higher_is_better": True,
            "scaling": "strong" if a.shard == "tile" else "weak",     # total work fixed as N grows (tile sharding, the default) vs per-GPU work fixed (sample sharding)
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.scene} scene {w}x{h} spp=1, {a.bounces} bounces, closest-hit + shadow rays counted on device",
                       "scene_sources": scene.describe(), "triangles": int(sst.triangles), "bvh_nodes": int(sst.bvh_nodes), "wide_layout": int(sst.wide_layout), "wide_depth": int(sst.wide_depth),
                       "scene_bytes": int(sst.scene_bytes), "scene_commit_wall_ms": round(r.device_scene.commit_wall_ms, 3),
                       "bvh_build_ms": round(sst.build_ms, 3), "bvh_build_mtris_per_s": round(sst.triangles / max(sst.build_ms, 1e-6) / 1e3, 1), "sah_cost": round(sst.sah_cost, 3),
                       "rays_per_frame": {"closest": closest / steps_total, "shadow": shadow / steps_total, "primary": primary / steps_total},
                       "shard": a.shard if world > 1 else "none", "frames_total": steps_total,
                       "frame_batch": frame_batch, "passes_of_timed_draw": plan["timed_passes"], "frames_in_flight": int(r.get_option("frames_in_flight")), "lanes_used": int(r.get_option("lanes_used")),
                       "lane_bytes": int(r.get_option("lane_bytes")), "persistent_traversal": int(r.get_option("persistent")),
                       "ms_per_step_is": "wall time of the timed region / steps with passes of frame_batch frames overlapped on frames_in_flight streams (inverse throughput); the per-frame device time is latency.ms_per_frame",
                       "device": r.ctx.device_name},
            # filled below from the serialised-pass leg (the kernel alone on the chip); without that leg (--no-latency, N > 1) it holds the figure under overlap and says so
            "roofline": {"bound": "hbm", "kernel": kernel_label,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": None, "traffic_source": None, "traffic_note": prof_note,
                         "algorithmic_bytes_per_launch": round(bytes_per_launch), "rays_per_launch": round(rays_per_launch, 1), "frames_per_launch": round(a.steps / passes, 3),
                         "bytes_per_closest_ray": BYTES_PER_CLOSEST_RAY, "bytes_per_shadow_ray": BYTES_PER_SHADOW_RAY,
                         "avg_launch_ms": round(avg_ms, 4), "launches_timed": t_n,
                         "regime": "UNDER OVERLAP (no serialised leg in this run): launches of the timed region, up to frames_in_flight passes sharing the chip — a launch's duration is then mostly time-slicing, not the kernel",
                         "under_overlap": {"avg_launch_ms": round(avg_ms, 4), "launches_timed": t_n, "frames_per_launch": round(a.steps / passes, 3), "algorithmic_bytes_per_launch": round(bytes_per_launch),
                                           "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5), "unit": "GB/s",
                                           "all_kernels_avg_launch_ms": {k: round(ms / n, 4) for k, (ms, n) in kt.items() if n},
                                           # the committed rocprofv3 --kernel-trace --stats of the driver's command shape (20 steps, 6 streams, passes of 7 + 7 + 6 frames)
                                           "avg_launch_ms_rocprof_driver_command": (round((prof.get("kernel_avg_us_driver") or {}).get("k_trace_mixed_wide_persist", 0.0) / 1e3, 4) or None) if prof_applies else None,
                                           "note": "the dominant kernel's own start/stop events inside the timed region: launches of up to frames_in_flight passes overlap on the GPU, so this is a duration under time-slicing (what rocprofv3 --kernel-trace of the same command reports), not the kernel alone"},
                         "frame": {"bytes_alg_per_frame": round(frame_bytes), "achieved": round(frame_gbs, 2), "frac": round(frame_gbs / HBM_PEAK_GBS, 5), "unit": "GB/s",
                                   "note": "SURVEY §8(d): (36 B/pixel + 96 B/closest ray + 72 B/shadow ray + one read of the scene) / t_frame"}},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if steady is not None:
            out["steady"] = steady
        if a.dump_accum:
            np.save(a.dump_accum, dump_img if world > 1 else r.accumulation())
        if a.png:
            if world > 1:
                r.write_accum_from(sr.buffer.data_ptr(), w * h * 16)      # show the assembled image, not this rank's shard
            mrt.save_png(a.png, r.tonemapped())
        if world == 1 and not a.no_latency:
            # ceilings calibrated on this chip, now: v_fma_f32 issue rate with every SIMD full, divergent-gather rate from a table of the scene's size
            import ctypes as C
            cal = (C.c_double * 5)()
            mrt._ffi.check(mrt.lib.mrt_debug_calibrate(r.ctx.handle, 128 << 20, cal))
            out["calibration"] = {"v_fma_f32_Ginst_per_s": round(cal[0] / 1e9, 1), "v_pk_fma_f32_Ginst_per_s": round(cal[1] / 1e9, 1), "shader_clock_GHz_under_fma_load": round(cal[4] / 1e9, 3),
                                  "gather16_GBps_128MiB_table": round(cal[2] / 1e9, 1), "gather80_GBps_128MiB_table": round(cal[3] / 1e9, 1),
                                  "note": "only fp32 add/mul/fma and and/or/xor/mov/lshr issue at this rate on gfx950; min/max, conversions, compares, shifts left, bit-field and 24-bit integer ops take ~1.8x as long, rcp/sqrt 3.5x (tools/valu_rates.hip, profiles/r02_valu_rates.json)"}
        if world == 1 and not a.no_latency:
            out["latency"] = latency_leg(mrt, r, scene, w, h, a.bounces, opts)
            out["ms_per_frame"] = out["latency"]["ms_per_frame"]      # SURVEY §8(d)'s ms/frame (one frame alone on the GPU), next to ms_per_step (inverse throughput)
            t4 = out["latency"]["kernel_ms_serialised_pass"].get("trace")
            if t4 and fused:
                # THE roofline figure: the dominant kernel alone on the chip at the default pass size — one stream, passes of PASS_FRAMES frames, its own start/stop events.
                # Algorithmic bytes of one such launch = this run's device-counted rays per frame x PASS_FRAMES / max_bounces launches per pass.
                b4 = (BYTES_PER_CLOSEST_RAY * (closest - primary) + BYTES_PER_SHADOW_RAY * shadow) / steps_total * PASS_FRAMES / a.bounces
                n4 = out["latency"]["launches_serialised_pass"].get("trace", 0)
                R = out["roofline"]
                R.update({"achieved": round(b4 / (t4 * 1e-3) / 1e9, 2), "frac": round(b4 / (t4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                          "avg_launch_ms": t4, "launches_timed": n4, "frames_per_launch": PASS_FRAMES, "algorithmic_bytes_per_launch": round(b4),
                          "rays_per_launch": round(((closest - primary) + shadow) / steps_total * PASS_FRAMES / a.bounces, 1),
                          "kernel_time_per_frame_ms": round(t4 * a.bounces / PASS_FRAMES, 4),
                          "regime": f"the kernel ALONE: one stream, passes of {PASS_FRAMES} frames (the default pass size), nothing else on the chip; HIP start/stop events of its own launches, taken in this run after the timed region.  kernel_time_per_frame_ms <= ms_per_step is the consistency check; under_overlap has the same kernel inside the timed region",
                          "avg_launch_ms_rocprof_serialised_pass": (round((prof.get("kernel_avg_us_serial_pass") or {}).get("k_trace_mixed_wide_persist", 0.0) / 1e3, 4) or None) if prof_applies else None,
                          "avg_launch_ms_rocprof_serialised_one_frame": (round((prof.get("kernel_avg_us_serial") or {}).get("k_trace_mixed_wide_stream", 0.0) / 1e3, 4) or None) if prof_applies else None})
                tw = out["latency"].get("kernel_ms_serialised_pass_whole_chip", {}).get("trace")
                if tw:
                    R["whole_chip"] = {"avg_launch_ms": tw, "achieved": round(b4 / (tw * 1e-3) / 1e9, 2), "frac": round(b4 / (tw * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "unit": "GB/s", "waves": out["latency"]["wave_slots"],
                                       "note": "the same serialised 8-frame launches on ALL the wave slots of the chip (renderer option wave_slots set): the kernel alone with nothing to leave room for.  The headline figure above keeps the launch shape of the timed region — a long call runs its pulling launches on half the slots so that the other passes' launches overlap (+2...+5 % on the frame) — which is also the shape of the committed rocprofv3 traces"}
                if prof_applies and traffic_p and prof.get("pmc_frames_per_dispatch", PASS_FRAMES) == PASS_FRAMES:
                    # the counters were collected on exactly these launches (serialised passes of PASS_FRAMES frames, tools/collect_profiles.sh): per launch, no scaling
                    R.update({"traffic": round(traffic_p), "traffic_source": prof.get("_source"),
                              "counters": {"source": prof.get("_source"), "valu_busy_pct": (prof.get("valu_busy_pct") or {}).get("k_trace_mixed_wide_persist"),
                                           "valu_lane_utilization_pct": (prof.get("valu_lane_utilization_pct") or {}).get("k_trace_mixed_wide_persist"),
                                           "note": "rocprofv3 --pmc passes over the same serialised 8-frame launches (VALUBusy, VALUUtilization; FETCH_SIZE x 2 + WRITE_SIZE per MI355X_MICROARCH.md): measured, per dispatch of this kernel"}})
            # the dominant kernel ALONE: one-frame launches on one stream (the latency leg's own start/stop events), algorithmic bytes of one frame's
            # bounce + shadow rays spread over its max_bounces launches — the kernel-level figure that profiles/r02_kernel_stats_serial.csv reproduces
            t_ser = out["latency"]["kernel_ms_serialised"].get("trace")
            if t_ser and fused:
                b_ser = (BYTES_PER_CLOSEST_RAY * (closest - primary) + BYTES_PER_SHADOW_RAY * shadow) / steps_total / a.bounces
                out["roofline"]["serialised_one_frame_launch"] = {"avg_launch_ms": t_ser, "algorithmic_bytes_per_launch": round(b_ser), "achieved": round(b_ser / (t_ser * 1e-3) / 1e9, 2),
                                                                  "frac": round(b_ser / (t_ser * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "unit": "GB/s"}
        if world == 1 and a.bounces > 1 and not a.no_strict:
            # the strict "primary + shadow" figure: the same renderer with max_bounces = 1
            r.set_option("max_bounces", 1); r.frameIndex = 0
            r.draw(a.warmup, wait=True); r.reset_stats(); torch.cuda.synchronize()
            t1 = time.perf_counter(); r.draw(a.steps, wait=True); dt1 = time.perf_counter() - t1
            s1 = r.stats
            out["strict_primary_plus_shadow"] = {"value": round((s1.closest_rays + s1.shadow_rays) / dt1 / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round(dt1 * 1e3 / a.steps, 4),
                                                 "rays_per_frame": {"primary": s1.closest_rays / a.steps, "shadow": s1.shadow_rays / a.steps}, "max_bounces": 1}
            r.set_option("max_bounces", a.bounces)
        cb = None
        if world == 1 and not a.no_cpu_baseline:
            try:
                cb, ref = cpu_baseline(mrt, scene, w, h, a.bounces, a.cpu_threads, two_level=any(kv.startswith("instancing=1") for kv in a.sopt))
                if oracle_note: cb["note"] = oracle_note
            except Exception as e:                 # the GPU result is printed whatever happens to the (reported, non-target) CPU leg
                out["cpu_baseline"] = None; out["cpu_baseline_note"] = (oracle_note or "") + f" cpu_baseline leg failed: {type(e).__name__}: {e}"
        if cb is not None:
            out["cpu_baseline"] = cb
            # parity of frame 0 against the oracle, same seeds (informational; the gates are tests/ -m gpu)
            r0 = mrt.Renderer((w, h), scene, ctx=r.ctx, seed=1, max_bounces=a.bounces, scene_options=opts)
            r0.draw(1, wait=True)
            g = r0.accumulation(); r0.close()
            d = np.abs(g[..., :3].astype(np.float64) - ref[..., :3])
            out["parity"] = {"bit_exact_pixels": float((g.view(np.uint32) == ref.view(np.uint32)).all(-1).mean()),
                             "rmse": float(np.sqrt((d ** 2).sum(-1).mean())), "within_1e-3": float((d.max(-1) <= 1e-3).mean())}
        elif "cpu_baseline" not in out:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    sr.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
