"""The small completions of the C-ABI boundary: the Uniforms block as an input (ShaderTypes.h:89-97, Renderer.swift:216-229), completion
without blocking (Renderer.swift:285-287), the header as C99 and as C++17, and one HIP runtime per process whatever the import order."""
import ctypes as C
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "mrt_abi.h")


@pytest.mark.parametrize("compiler,std", [("gcc", "-std=c99"), ("gcc", "-std=c11"), ("g++", "-std=c++17"), ("g++", "-std=c++11")])
def test_header_is_plain_c_and_cxx(tmp_path, compiler, std):
    """include/mrt_abi.h must be consumable by a C99 host and by a C++ host alike (no C++ types, no extensions)."""
    if shutil.which(compiler) is None:
        pytest.skip(compiler + " not installed")
    src = tmp_path / ("t.c" if compiler == "gcc" else "t.cpp")
    src.write_text('#include "mrt_abi.h"\n#include "mrt_debug.h"\n'
                   "int main(void) { MRTUniforms u; MRTLight l; MRTRenderStats s; (void)u; (void)l; (void)s;\n"
                   "  return (sizeof(MRTCamera) == 64 && sizeof(MRTLight) == 128 && sizeof(MRTUniforms) == 96 && sizeof(MRTMaterial) == 64 && sizeof(MRTRay) == 32 && sizeof(MRTIntersection) == 32) ? 0 : 1; }\n")
    exe = tmp_path / "t"
    subprocess.check_call([compiler, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_group_partition_is_a_partition():
    """Host logic of the device group (mrt_group_*): tile t belongs to rank t % n — every pixel has exactly one owner, ragged edges included."""
    for w, h, n in [(1920, 1080, 8), (37, 21, 3), (8, 8, 2), (1, 1, 4), (3840, 2160, 5)]:
        tx, ty = (w + 7) // 8, (h + 7) // 8
        owner = (np.arange(ty)[:, None] * tx + np.arange(tx)[None, :]) % n
        px = np.repeat(np.repeat(owner, 8, 0), 8, 1)[:h, :w]
        counts = np.bincount(px.ravel(), minlength=n)
        assert counts.sum() == w * h
        # ranks differ by at most one tile row's worth of pixels per tile column: interleaving balances the load
        full_tiles = (w // 8) * (h // 8)
        if full_tiles >= 8 * n:
            assert counts.max() - counts.min() <= 64 * (tx + ty)


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_uniforms_block_as_input(mrt, orc, gpu_ctx):
    from test_gpu_parity import assert_parity, oracle_render
    w, h = 96, 64
    sc = mrt.CornellScene((w, h))
    second = mrt.Light.pointLight([0.3, 1.2, 0.4], [2, 3, 1])
    sc.lights = [mrt.Scene.setupLight(), second]
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    u = r.uniforms
    assert (u.width, u.height, u.blocksWide, u.frameIndex, u.lightCount) == (w, h, (w + 7) // 8, 0, 2)
    assert bytes(u.camera) == bytes(sc.camera)
    # lightCount = 1: the kernels sample only the first light (Raytracing.metal:273, :335) == a scene that holds only that light
    u.lightCount = 1
    r.uniforms = u
    r.draw(2, wait=True)
    one = mrt.CornellScene((w, h)); one.lights = [mrt.Scene.setupLight()]
    ref, cnt = oracle_render(orc, mrt, one, w, h, 2)
    assert_parity(r.accumulation(), ref, exact_frac=1.0)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    # frameIndex through the block == mrt_renderer_set_frame_index; all lights again
    u = r.uniforms
    assert u.frameIndex == 2 and u.lightCount == 1
    u.lightCount = 2; u.frameIndex = 5
    r.uniforms = u
    assert r.frameIndex == 5
    # a new size resizes (new targets, new seeds: Renderer.swift:353-356) and then applies frameIndex; a moved camera is taken
    u.width, u.height, u.frameIndex = 50, 19, 0
    cam = mrt.Scene.setupCamera((50, 19)); cam.position.x += 0.25
    u.camera = cam
    r.uniforms = u
    r.draw(1, wait=True)
    sc2 = mrt.CornellScene((50, 19)); sc2.lights = sc.lights; sc2.camera = cam
    ref2, _ = oracle_render(orc, mrt, sc2, 50, 19, 1)
    assert_parity(r.accumulation(), ref2, exact_frac=1.0)
    # refused: lightCount outside [1, lights of the scene]
    for bad in (0, 3):
        u = r.uniforms; u.lightCount = bad
        with pytest.raises(mrt.MRTError):
            r.uniforms = u
    r.close()


@pytest.mark.gpu
def test_frames_completed_never_blocks_and_ends_at_the_total(mrt, gpu_ctx):
    w, h = 640, 360
    r = mrt.Renderer((w, h), mrt.CornellScene((w, h)), ctx=gpu_ctx)
    assert r.framesCompleted == 0
    assert r.get_option("frame_batch") == 32          # the default, by image size: 640 x 360 is a ninth of 1920 x 1080 (8 there, at most 32)
    r.set_option("frame_batch", 8)                    # this test counts passes of eight
    r.draw(3, wait=True)
    assert r.framesCompleted == 3 == r.stats.frames
    n = 400
    t0 = time.perf_counter()
    r.draw(n)                                   # enqueues and returns (Renderer.swift:284-351 is asynchronous)
    seen = [r.framesCompleted]
    t_poll = time.perf_counter() - t0
    while seen[-1] < 3 + n and time.perf_counter() - t0 < 60:
        seen.append(r.framesCompleted)
    assert seen[-1] == 3 + n
    assert all(b >= a for a, b in zip(seen, seen[1:]))                      # monotonic
    assert all((s - 3) % 8 == 0 or s == 3 + n for s in seen if s > 3)       # reported per pass of frame_batch = 8 frames (400 = 50 x 8)
    # the last passes of a draw (one per pass in flight: 6 of these 50) are accumulated in ONE launch after the call's last traversal launch, so their frames are
    # reported together: no value strictly between 3 + 44 x 8 and the total can ever be seen (include/mrt_abi.h mrt_renderer_frames_completed)
    lanes = int(r.get_option("lanes_used"))
    assert lanes >= 1 and not any(3 + (50 - lanes) * 8 < s < 3 + n for s in seen), (lanes, sorted(set(seen))[-4:])
    assert seen[0] < 3 + n, f"the first poll, {t_poll * 1e3:.1f} ms after the call returned, already saw every frame: render() blocked?"
    r.wait()
    assert r.framesCompleted == 3 + n
    r.reset_stats()
    assert r.framesCompleted == 0
    r.drawableSizeWillChange((64, 64))
    r.draw(2)
    r.wait()
    assert r.framesCompleted == 2
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["library_first", "torch_first"])
def test_one_hip_runtime_whatever_the_import_order(order):
    """libmrt_hip.so and PyTorch must share ONE HIP runtime: importing the package before torch used to leave torch with "No HIP GPUs"."""
    first, second = ("import metal_raytracing_amd as m", "import torch") if order == "library_first" else ("import torch", "import metal_raytracing_amd as m")
    code = (f"import sys; sys.path.insert(0, {ROOT!r})\n{first}\n{second}\n"
            "assert torch.cuda.is_available(), 'torch lost the GPU'\n"
            "x = torch.ones(1024, device='cuda').sum().item(); assert x == 1024.0\n"
            "ctx = m.Context(0); r = m.Renderer((32, 32), m.CornellScene((32, 32)), ctx=ctx); r.draw(1, wait=True); a = r.accumulation(); r.close(); ctx.close()\n"
            "import numpy as np; assert np.isfinite(a).all() and a[..., :3].max() > 0\n"
            "maps = open('/proc/self/maps').read(); hips = {l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}\n"
            "assert len(hips) == 1, hips\nprint('ok', hips)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
