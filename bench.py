#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: Mrays/s (closest-hit + shadow rays) and ms/frame,
DragonScene 1920x1080 spp=1 (configs[1]).

A "step" is one frame (one pass of the hot path over all pixels at 1 spp): primary rays, then per bounce
shade / trace (closest hit for the bounce rays + any hit for the shadow rays), then accumulate —
`Renderer.draw(in:)` of the reference (Renderer.swift:284-351).  Inputs (scene, BVH, seeds) are resident in HBM
before the timed region.  The renderer carries the K steps in batches of `frame_batch` frames on
`frames_in_flight` HIP streams (the reference keeps 3 frames in flight, Renderer.swift:33); every frame is
rendered in full and the running average is applied in frame order.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: one process per GPU, scene + BVH replicated.  Default `--shard sample`: every rank renders full frames of a
disjoint Halton index range (per-GPU work fixed → "weak" scaling); `--shard tile`: the image is sharded by 8x8
screen tile (tile_id % N == rank, total work fixed → "strong").  Either way the K frames are accumulated locally and
ONE RCCL reduce of the RGBA32F radiance buffer assembles the image on rank 0 inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # one hardware queue per frame in flight (before HIP initialises)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_CLOSEST_RAY = 96     # ray 2x32 B + hit 2x16 B
BYTES_PER_SHADOW_RAY = 72


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "dragon_irregular", "cornell", "dragon4", "garden"])
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--shard", default="sample", choices=["tile", "sample"], help="N > 1: sample-index sharding (weak scaling, default) or 8x8 screen-tile sharding (strong scaling)")
    ap.add_argument("--builder", type=int, default=None)
    ap.add_argument("--opt", action="append", default=[], help="renderer option key=value (repeatable)")
    ap.add_argument("--sopt", action="append", default=[], help="scene (BVH build) option key=value (repeatable)")
    ap.add_argument("--frames-in-flight", type=int, default=None, help="Renderer.maxFramesInFlight (default 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="skip the extra max_bounces=1 (primary + shadow only) measurement")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal on one GPU box (ranks share GPUs, reduce on host)")
    ap.add_argument("--png", default=None, help="write the tonemapped image here (rank 0)")
    return ap.parse_args()


def measured_traffic():
    """HBM bytes per traversal launch (call-weighted over k_trace_*) from the rocprofv3 PMC passes kept under profiles/ (tools/collect_profiles.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for
    gfx950; gather widths are uncalibrated, so this is an upper estimate).  None if no profile is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None, None, None, None
    try:
        d = json.load(open(files[-1]))
        return d["trace_launch_hbm_bytes"], os.path.relpath(files[-1], ROOT), d.get("trace_launch_avg_us"), d.get("valu_wave_insts_per_frame")
    except Exception:
        return None, None, None, None


def cpu_baseline(mrt, scene, w, h, bounces, threads):
    """The oracle (CPU restatement, kind 'port') on the GPU box's host cores: one full frame of the
    same workload, same seeds.  Reported, never the thing shipped."""
    import oracle as O
    O.build_oracle()
    threads = threads or min(os.cpu_count() or 1, 16)     # the GPU box's CPU share for one GPU is 16 cores
    osc = O.OracleScene(mrt.flatten_scene(scene), scene.lights)
    r = O.OracleRenderer(osc, w, h, seed=1, max_bounces=bounces, camera=scene.camera)
    t0 = time.perf_counter()
    r.render(1, threads=threads)
    dt = time.perf_counter() - t0
    closest, shadow = r.counters()
    img = r.accumulation()
    return {"value": (closest + shadow) / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"1 full frame of the same workload ({w}x{h} spp=1, {bounces} bounces, {closest + shadow} rays) in {dt:.2f} s",
            "ms_per_frame": dt * 1e3}, img


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        a.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = max(1, torch.cuda.device_count())
        local_rank = local_rank % ndev                      # rehearsal with more ranks than GPUs (gloo only)
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    elif a.gpus > 1:
        sys.exit("launch with torch.distributed.run for --gpus > 1")

    import metal_raytracing_amd as mrt
    from metal_raytracing_amd.distributed import reduce_accumulation
    w, h = a.width, a.height
    scene = mrt.SCENES[a.scene]((w, h))
    opts = {} if a.builder is None else {"builder": a.builder}
    for kv in a.sopt:
        k, v = kv.split("="); opts[k] = float(v)
    r = mrt.Renderer((w, h), scene, device=local_rank, seed=1, max_bounces=a.bounces, scene_options=opts)
    sst = r.device_scene.stats
    for kv in a.opt:
        k, v = kv.split("="); r.set_option(k, float(v))
    if a.frames_in_flight is not None:
        r.set_option("frames_in_flight", a.frames_in_flight)
    if world > 1:
        if a.shard == "tile":
            r.set_shard(rank, world)
            if a.frames_in_flight is None:
                r.set_option("frames_in_flight", 8)     # 1/N of the pixels per frame: more frames in flight to cover the per-kernel tails
        else:
            r.set_option("sample_offset", rank * (a.warmup + a.steps))

    def sync():
        r.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    npix = w * h
    accum_t = torch.zeros((h, w, 4), dtype=torch.float32, device=f"cuda:{local_rank}") if world > 1 else None

    # ---- warmup
    r.draw(a.warmup, wait=True)
    def reduce_image():
        r.copy_accum_to(accum_t.data_ptr(), npix * 16); r.wait()
        if a.dist_backend == "nccl":
            reduce_accumulation(accum_t, a.shard, dst=0)      # the ONE collective per output image (RCCL over xGMI)
        else:
            host = accum_t.cpu(); reduce_accumulation(host, a.shard, dst=0); accum_t.copy_(host)
    if world > 1:
        reduce_image()
    r.reset_stats()
    sync()
    # ---- timed region: exactly K steps (+ the one reduce of the output image for N > 1)
    t0 = time.perf_counter()
    done = 0
    ext_ms, ext_launches = 0.0, 0
    # all K steps are enqueued at once: the renderer carries them in batches of `frame_batch` frames on `frames_in_flight` streams;
    # its first 512 traversal launches carry start/stop events (the live roofline measurement)
    r.draw(a.steps)
    r.wait()
    st = r.stats
    ext_ms += st.ms_extend_last; ext_launches += st.extend_launches_last
    if world > 1:
        reduce_image()
    sync()
    dt = time.perf_counter() - t0
    st = r.stats
    rays = torch.tensor([st.closest_rays, st.shadow_rays, st.primary_rays], dtype=torch.float64)
    tmax = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        if a.dist_backend == "nccl":
            rays = rays.cuda(); tmax = tmax.cuda()
        dist.all_reduce(rays); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rays = rays.cpu(); tmax = tmax.cpu()
    dt = float(tmax[0])
    closest, shadow, primary = (float(x) for x in rays)
    steps_total = a.steps * (world if a.shard == "sample" and world > 1 else 1)

    if rank == 0:
        value = (closest + shadow) / dt / 1e6
        # dominant kernels: the traversal launches (k_trace_primary + k_trace_mixed; k_extend in the unfused pipeline).
        # Algorithmic bytes per launch = (96 B x closest-hit rays + 72 B x shadow rays) / traversal launches;
        # the fused launches carry both kinds, the unfused k_extend only the 96-B rays.
        fused = r.get_option("fused") != 0 and r.get_option("wide") == 0
        frame_batch = int(r.get_option("frame_batch")) if fused else 1
        passes = (a.steps + frame_batch - 1) // frame_batch              # one pass of the pipeline = frame_batch frames
        launches_per_frame = (a.bounces + 1 if fused else a.bounces) * passes / a.steps
        traced_bytes = BYTES_PER_CLOSEST_RAY * st.closest_rays + (BYTES_PER_SHADOW_RAY * st.shadow_rays if fused else 0)
        bytes_per_launch = traced_bytes / (a.steps * launches_per_frame)
        rays_per_launch = (st.closest_rays + (st.shadow_rays if fused else 0)) / (a.steps * launches_per_frame)
        avg_ms = ext_ms / max(1, ext_launches)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        frame_bytes = st.bytes_alg / max(1, st.frames)
        traffic, traffic_src, prof_avg_us, valu_insts = measured_traffic()
        out = {
            "metric": "Mrays/sec (primary+shadow) and ms/frame, DragonScene 1920x1080 spp=1" if (a.scene, w, h) == ("dragon", 1920, 1080) else f"Mrays/sec (closest+shadow), {a.scene} {w}x{h} spp=1",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt * 1e3 / a.steps, 4), "higher_is_better": True,
            "scaling": "weak" if a.shard == "sample" else "strong",     # per-GPU work fixed as N grows (sample sharding, the default) vs total work fixed (tile sharding)
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.scene} scene {w}x{h} spp=1, {a.bounces} bounces, closest-hit + shadow rays counted on device",
                       "scene_sources": scene.describe(), "triangles": int(sst.triangles), "bvh_nodes": int(sst.bvh_nodes),
                       "bvh_build_ms": round(sst.build_ms, 3), "sah_cost": round(sst.sah_cost, 3),
                       "rays_per_frame": {"closest": closest / steps_total, "shadow": shadow / steps_total, "primary": primary / steps_total},
                       "shard": a.shard if world > 1 else "none", "frames_total": steps_total,
                       "frame_batch": frame_batch, "frames_in_flight": int(r.get_option("frames_in_flight")),
                       "frame_bytes_alg": frame_bytes, "frame_alg_GBps": round(frame_bytes * st.frames / dt / 1e9, 2),
                       "device": r.ctx.device_name},
            "roofline": {"bound": "hbm", "kernel": "k_trace_primary+k_trace_mixed" if fused else "k_extend", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(bytes_per_launch),
                         "bytes_per_closest_ray": BYTES_PER_CLOSEST_RAY, "bytes_per_shadow_ray": BYTES_PER_SHADOW_RAY, "rays_per_launch": round(rays_per_launch, 1), "avg_launch_ms": round(avg_ms, 4),
                         # the live figure spans end-of-previous-command .. end-of-kernel on the launching stream, i.e. it includes the dispatch gap,
                         # which grows with the frames in flight; the committed rocprofv3 kernel time of the same command:
                         "avg_launch_ms_rocprof": round(prof_avg_us / 1e3, 4) if prof_avg_us else None,
                         "launches_timed": ext_launches},
        }
        if valu_insts and world == 1 and (a.scene, w, h, a.bounces) == ("dragon", 1920, 1080, 3):
            # what actually bounds the frame: VALU issue.  Wave-instructions per frame from the committed SQ_INSTS_VALU pass
            # (a property of the workload) over the NOMINAL issue capacity of the timed frame: 256 CUs x 4 SIMDs x one wave64 instruction per
            # 4 cycles at 2.4 GHz.  The counter also counts instructions that retire early with an empty EXEC mask, so frac can read > 1.
            VALU_PEAK = 256 * 2.4e9
            out["valu_issue"] = {"wave_insts_per_frame": round(valu_insts), "achieved_Ginst_per_s": round(valu_insts / (dt / a.steps) / 1e9, 1),
                                 "nominal_peak_Ginst_per_s": VALU_PEAK / 1e9, "frac": round(valu_insts / (dt / a.steps) / VALU_PEAK, 4), "clock_GHz_assumed": 2.4,
                                 "source": traffic_src}
        if a.png:
            if world > 1:
                r.write_accum_from(accum_t.data_ptr(), npix * 16)      # show the assembled image, not this rank's shard
            mrt.save_png(a.png, r.tonemapped())
        if world == 1 and a.bounces > 1 and not a.no_strict:
            # the strict "primary + shadow" figure: the same renderer with max_bounces = 1
            r.set_option("max_bounces", 1); r.frameIndex = 0
            r.draw(a.warmup, wait=True); r.reset_stats(); torch.cuda.synchronize()
            t1 = time.perf_counter(); r.draw(a.steps, wait=True); dt1 = time.perf_counter() - t1
            s1 = r.stats
            out["strict_primary_plus_shadow"] = {"value": round((s1.closest_rays + s1.shadow_rays) / dt1 / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round(dt1 * 1e3 / a.steps, 4),
                                                 "rays_per_frame": {"primary": s1.closest_rays / a.steps, "shadow": s1.shadow_rays / a.steps}, "max_bounces": 1}
            r.set_option("max_bounces", a.bounces)
        if world == 1 and not a.no_cpu_baseline:
            cb, ref = cpu_baseline(mrt, scene, w, h, a.bounces, a.cpu_threads)
            out["cpu_baseline"] = cb
            # parity of frame 0 against the oracle, same seeds (informational; the gates are tests/ -m gpu)
            r0 = mrt.Renderer((w, h), scene, ctx=r.ctx, seed=1, max_bounces=a.bounces, scene_options=opts)
            r0.draw(1, wait=True)
            g = r0.accumulation(); r0.close()
            d = np.abs(g[..., :3].astype(np.float64) - ref[..., :3])
            out["parity"] = {"bit_exact_pixels": float((g.view(np.uint32) == ref.view(np.uint32)).all(-1).mean()),
                             "rmse": float(np.sqrt((d ** 2).sum(-1).mean())), "within_1e-3": float((d.max(-1) <= 1e-3).mean())}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
