"""bench.py prints ONE JSON line with the driver's contract keys plus the `roofline` and `cpu_baseline` objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_plan_of_every_rank_count_matches_design_section_7():
    """bench.py's N > 1 argument / environment handling without a GPU: what torch.distributed.run's variables make of --gpus, which device a rank binds, the
    frame_batch a tile-sharded rank gets and the passes its draws split into — the per-rank configuration DESIGN.md §7 states (and SCALE records are read against)."""
    sys.path.insert(0, ROOT)
    import bench
    from metal_raytracing_amd.distributed import pass_sizes, shard_frame_batch
    # the driver's command: --steps 20 --warmup 5 on N = 1, 2, 4, 8 (25 frames in all: a pass takes at most a third of them, never fewer than 8)
    for n in (1, 2, 4, 8):
        env = {} if n == 1 else {"WORLD_SIZE": str(n), "RANK": str(n - 1), "LOCAL_RANK": str(n - 1)}
        p = bench.launch_plan(1 if n == 1 else n, 20, 5, "tile", env, ndev=8)
        assert p["world"] == n and p["gpus"] == n and p["device"] == n - 1 and p["mode"] == ("single" if n == 1 else "ranks")
        assert p["frame_batch"] == 8 and p["timed_passes"] == [7, 7, 6] and p["warmup_passes"] == [5], (n, p)
    # a long run (240 + 24 frames): 8 x N frames per pass, at most 32; the timed draw in passes of equal size
    for n, fb, passes in ((1, 8, [8] * 30), (2, 16, [16] * 15), (4, 32, [30] * 8), (8, 32, [30] * 8)):
        env = {} if n == 1 else {"WORLD_SIZE": str(n), "RANK": "0", "LOCAL_RANK": "0"}
        p = bench.launch_plan(n, 240, 24, "tile", env, ndev=8)
        assert p["frame_batch"] == fb and p["timed_passes"] == passes and sum(p["timed_passes"]) == 240, (n, p)
    # WORLD_SIZE wins over a stale --gpus; more ranks than devices (a gloo rehearsal on one box) wrap around; sample sharding keeps the 1-GPU pass
    p = bench.launch_plan(1, 20, 5, "tile", {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}, ndev=1)
    assert p["gpus"] == 4 and p["device"] == 0 and p["mode"] == "ranks"
    assert bench.launch_plan(8, 20, 5, "sample", {"WORLD_SIZE": "8", "RANK": "1", "LOCAL_RANK": "1"}, ndev=8)["frame_batch"] == 8
    # --gpus N without torch.distributed.run: one process, the C ABI's device group (min(32, 8 N) per rank; every draw capped by Renderer::render itself)
    p = bench.launch_plan(8, 20, 5, "tile", {}, ndev=8)
    assert p["mode"] == "group" and p["world"] == 1 and p["gpus"] == 8 and p["frame_batch"] == 32 and p["timed_passes"] == [7, 7, 6]
    # the rule itself
    assert [shard_frame_batch(n) for n in (1, 2, 3, 4, 8, 16)] == [8, 16, 24, 32, 32, 32]
    assert shard_frame_batch(8, frames_total=25) == 8 and shard_frame_batch(8, frames_total=60) == 20 and shard_frame_batch(2, frames_total=1000) == 16
    assert pass_sizes(20, 8) == [7, 7, 6] and pass_sizes(20, 32) == [7, 7, 6] and pass_sizes(16, 8) == [8, 8] and pass_sizes(1, 8) == [1] and pass_sizes(48, 8) == [8] * 6
    assert pass_sizes(100, 32) == [25] * 4 and pass_sizes(5, 1) == [1] * 5
    # the library's default frames per pass (frame_batch = 0): by image size, eight 1080p frames' worth of pixel-frames, 8 ... 32
    from metal_raytracing_amd.distributed import auto_frame_batch
    assert [auto_frame_batch(w * h) for w, h in ((3840, 2160), (1920, 1080), (1600, 900), (960, 540), (640, 360), (256, 256), (1, 1))] == [8, 8, 12, 32, 32, 32, 32]
    assert bench.launch_plan(1, 240, 24, "tile", {}, ndev=1, pixels=256 * 256)["timed_passes"] == [30] * 8 and bench.launch_plan(1, 20, 5, "tile", {}, ndev=1, pixels=256 * 256)["timed_passes"] == [7, 7, 6]


def _roofline_identities(d, full_size=False):
    """What makes `roofline` reproducible: frac x peak x avg_launch_ms IS the algorithmic bytes of one launch; those bytes are this run's device-counted rays per frame
    x frames per launch / launches per pass at SURVEY §8(d)'s 96 B / 72 B; and (full-size runs, where a launch fills the chip) the kernel's serialised time per frame
    fits inside the frame's inverse throughput."""
    r = d["roofline"]
    assert abs(r["frac"] * r["peak"] * 1e9 * r["avg_launch_ms"] * 1e-3 / r["algorithmic_bytes_per_launch"] - 1.0) < 2e-3, r
    rays = d["config"]["rays_per_frame"]
    bounces = 3
    want = (r["bytes_per_closest_ray"] * (rays["closest"] - rays["primary"]) + r["bytes_per_shadow_ray"] * rays["shadow"]) * r["frames_per_launch"] / bounces
    assert r["bytes_per_closest_ray"] == 96 and r["bytes_per_shadow_ray"] == 72 and abs(want / r["algorithmic_bytes_per_launch"] - 1.0) < 2e-3, (want, r["algorithmic_bytes_per_launch"])
    if "ALONE" in r["regime"]:
        assert abs(r["kernel_time_per_frame_ms"] - r["avg_launch_ms"] * bounces / r["frames_per_launch"]) < 1e-3
        if full_size:
            assert r["kernel_time_per_frame_ms"] <= d["ms_per_step"], (r["kernel_time_per_frame_ms"], d["ms_per_step"])


def _latest_profile_line(name):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + name)))
    return files[-1] if files else None


def test_committed_driver_command_line_is_reproducible_from_profiles():
    """The bench line of the driver's command (`--steps 20 --warmup 5`, full size) kept under profiles/ with the rocprofv3 summaries of the same tree: its roofline
    obeys the identities above, the kernel's serialised time per frame fits in ms_per_step, and — when the line carries the committed rocprofv3 average of the same
    serialised 8-frame launches — the live HIP-event figure agrees with it within 5 %."""
    f = _latest_profile_line("bench_driver_command.json")
    assert f, "profiles/rNN_bench_driver_command.json missing"
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    if "regime" not in r:
        pytest.skip(f"{os.path.basename(f)} predates the serialised-pass roofline (round 5)")
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and "1920x1080" in d["metric"]
    _roofline_identities(d, full_size=True)
    assert "ALONE" in r["regime"] and "under_overlap" in r and "valu_issue" not in d
    rp = r.get("avg_launch_ms_rocprof_serialised_pass")
    if rp:
        assert abs(r["avg_launch_ms"] / rp - 1.0) < 0.05, (r["avg_launch_ms"], rp)
        assert r["traffic"] and r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]          # counted HBM bytes of the same launches: never (much) below the algorithmic ones


@pytest.mark.gpu
def test_bench_json_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--width", "320", "--height", "180", "--cpu-threads", "4"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and "traffic" in r
    assert r["frame"]["frac"] > 0 and r["launches_timed"] > 0 and r["avg_launch_ms"] > 0
    _roofline_identities(d)
    # the headline figure is the kernel ALONE (serialised 8-frame passes: 5 draws x 2 passes x 3 bounces); the timed region's own launches sit in under_overlap
    assert "ALONE" in r["regime"] and r["frames_per_launch"] == 8 and r["launches_timed"] == 30 and r["avg_launch_ms"] == d["latency"]["kernel_ms_serialised_pass"]["trace"]
    u = r["under_overlap"]
    assert d["config"]["passes_of_timed_draw"] == [6] and u["launches_timed"] == 3 * len(d["config"]["passes_of_timed_draw"])      # the passes bench.py's launch_plan predicts are the ones the library ran (one traversal launch per bounce and pass)
    assert u["avg_launch_ms"] > 0 and abs(u["frac"] - u["achieved"] / r["peak"]) < 1e-4 and "valu_issue" not in d
    assert d["config"]["wide_layout"] == 1 and d["config"]["wide_depth"] >= 1 and d["config"]["scene_commit_wall_ms"] > 0
    lat = d["latency"]
    assert lat["ms_per_frame"] > 0 and lat["kernel_ms_serialised"]["trace"] > 0 and lat["reference_like_3_in_flight_ms_per_frame"] > 0
    assert d["calibration"]["v_fma_f32_Ginst_per_s"] > 100
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 4 and c["value"] > 0 and c["unit"] == "Mrays/s" and "sample" in c
    assert d["parity"]["bit_exact_pixels"] == 1.0 and d["parity"]["rmse"] == 0.0


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_assemble_the_one_rank_image(tmp_path):
    """bench.py's own N > 1 path (ShardedRenderer: tile sharding, one reduce per image) rehearsed with 2 ranks that share this
    box's GPU over gloo: the image rank 0 assembles must equal the 1-rank image bit for bit, and the JSON line must say
    n_gpus = 2, strong scaling.  So the first real 8-GPU run is not the first time that code executes."""
    import numpy as np
    common = ["--steps", "6", "--warmup", "2", "--width", "320", "--height", "180", "--no-cpu-baseline", "--no-strict", "--no-latency"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--dump-accum", one], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                         os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", *common, "--steady-steps", "12", "--dump-accum", two],
                        capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert p2.returncode == 0, p2.stderr[-3000:]
    d = json.loads([l for l in p2.stdout.strip().splitlines() if l.startswith("{")][-1])
    # the N > 1 line carries both regimes: the driver's steps (value) and a longer leg after them (steady) with its own per-rank times
    s = d["steady"]
    assert s["steps"] == 12 and s["value"] > 0 and s["ms_per_step"] > 0 and len(s["per_rank"]["ms_gpu_timed_draw"]) == 2 and "reduce" in d["per_rank"]["assemble"]
    # the compact assemble (every rank ships only its own tiles) gives the same image
    three = str(tmp_path / "three.npy")
    p3 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29543",
                         os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", *common, "--steady-steps", "0", "--assemble", "compact", "--dump-accum", three],
                        capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert p3.returncode == 0, p3.stderr[-3000:]
    d3 = json.loads([l for l in p3.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert "compact" in d3["per_rank"]["assemble"] and "steady" not in d3 and np.array_equal(np.load(one), np.load(three))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["shard"] == "tile" and d["value"] > 0
    pr = d["per_rank"]           # every rank's device time and its time in the reduce: what the first real SCALE record will be diagnosed with
    assert len(pr["ms_gpu_timed_draw"]) == 2 and min(pr["ms_gpu_timed_draw"]) > 0 and len(pr["gather_wall_ms"]) == 2 and max(pr["render_wall_ms"]) * 1e-3 <= d["ms_per_step"] * d["steps"] * 1e-3 + 1e-3
    d1 = json.loads([l for l in p1.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d1["config"]["rays_per_frame"] == d["config"]["rays_per_frame"]          # ray-count conservation across the shards
    a, b = np.load(one), np.load(two)
    assert a.shape == (180, 320, 4) and np.array_equal(a, b)


@pytest.mark.gpu
def test_bench_group_path_assembles_the_one_rank_image(tmp_path):
    """`bench.py --gpus 2` without torch.distributed.run drives the C ABI's device group (mrt_group_*); rehearsed here with a group that names
    this box's GPU twice: same image as one rank, n_gpus = 2, strong scaling, ray counts conserved."""
    import numpy as np
    common = ["--steps", "6", "--warmup", "2", "--width", "320", "--height", "180", "--no-cpu-baseline", "--no-strict", "--no-latency"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--dump-accum", one], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--group-devices", "0,0", *common, "--steady-steps", "12", "--assemble", "compact", "--dump-accum", two], capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert p2.returncode == 0, p2.stderr[-3000:]
    d = json.loads([l for l in p2.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["config"]["reduce_mode"] == 2 and "compact" in d["config"]["reduce"] and d["steady"]["steps"] == 12 and d["steady"]["value"] > 0
    d1 = json.loads([l for l in p1.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["shard"] == "tile" and d["value"] > 0 and "device group" in d["config"]["launch"]
    assert len(d["per_rank"]["ms_gpu_timed_draw"]) == 2 and min(d["per_rank"]["ms_gpu_timed_draw"]) > 0 and d["per_rank"]["gather_wall_ms"] >= 0
    assert d1["config"]["rays_per_frame"] == d["config"]["rays_per_frame"]
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a, b)


def test_bench_argparse_defaults():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    assert 'default=1)' in src.split('"--gpus"')[1][:60]
