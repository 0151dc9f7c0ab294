"""The C++ host-side mirror (include/mrt.hpp: Scene / DragonScene / Model / Mesh / Submesh / Renderer over
the C ABI) compiles against the header, fails loudly without a GPU, and on the GPU gives the same frame
as the Python mirror."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "dragon_scene")


def _build():
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "dragon_scene.cpp"),
                           "-L" + os.path.join(ROOT, "metal-raytracing_amd"), "-lmrt_hip", "-Wl,-rpath," + os.path.join(ROOT, "metal-raytracing_amd"), "-o", EXE])


def test_cpp_mirror_builds_and_fails_loudly_without_gpu():
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([EXE, "32", "32", "1"], capture_output=True, text=True, cwd=ROOT)
    assert p.returncode == 1 and "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_cpp_mirror_matches_python_mirror(mrt, gpu_ctx):
    _build()
    w, h, frames = 160, 90, 3
    p = subprocess.run([EXE, str(w), str(h), str(frames)], capture_output=True, text=True, cwd=ROOT)
    assert p.returncode == 0, p.stderr
    m = re.search(r"triangles=(\d+) frames=(\d+) frameIndex=(\d+) closest=(\d+) shadow=(\d+) ms=\S+ checksum=(\S+)", p.stdout)
    assert m, p.stdout
    r = mrt.Renderer((w, h), mrt.DragonScene((w, h)), ctx=gpu_ctx)
    r.draw(frames, wait=True)
    acc = r.accumulation(); st = r.stats
    assert int(m.group(1)) == 885194 and int(m.group(2)) == frames and int(m.group(3)) == frames
    assert (int(m.group(4)), int(m.group(5))) == (st.closest_rays, st.shadow_rays)
    checksum = float(np.sum(acc[..., 0].astype(np.float64) + acc[..., 1].astype(np.float64) + acc[..., 2].astype(np.float64)))
    # the C++ side sums in the same element order in double: identical up to printing precision
    assert abs(float(m.group(6)) - checksum) <= 1e-6 * max(1.0, abs(checksum))
    r.close()


@pytest.mark.gpu
def test_cpp_mirror_two_level_matches_python_two_level(mrt, gpu_ctx):
    """instancing through the C++ mirror (the two spheres of DragonScene become one BLAS x two instances) == the Python mirror's."""
    _build()
    w, h, frames = 128, 72, 2
    p = subprocess.run([EXE, str(w), str(h), str(frames), "-", "1"], capture_output=True, text=True, cwd=ROOT)
    assert p.returncode == 0, p.stderr
    m = re.search(r"triangles=(\d+) frames=(\d+) frameIndex=(\d+) closest=(\d+) shadow=(\d+) ms=\S+ checksum=(\S+)", p.stdout)
    assert m, p.stdout
    r = mrt.Renderer((w, h), mrt.DragonScene((w, h)), ctx=gpu_ctx, scene_options={"instancing": 1})
    r.draw(frames, wait=True)
    acc = r.accumulation(); st = r.stats
    assert (int(m.group(4)), int(m.group(5))) == (st.closest_rays, st.shadow_rays)
    checksum = float(np.sum(acc[..., 0].astype(np.float64) + acc[..., 1].astype(np.float64) + acc[..., 2].astype(np.float64)))
    assert abs(float(m.group(6)) - checksum) <= 1e-6 * max(1.0, abs(checksum))
    r.close()


@pytest.mark.gpu
def test_cpp_mirror_group_renderer_matches_single_device(mrt, gpu_ctx):
    """mrt::GroupRenderer (mrt_group_*) over a group that names device 0 twice == mrt::Renderer on that device."""
    _build()
    w, h, frames = 160, 90, 5
    one = subprocess.run([EXE, str(w), str(h), str(frames)], capture_output=True, text=True, cwd=ROOT)
    grp = subprocess.run([EXE, str(w), str(h), str(frames), "-", "0", "0,0"], capture_output=True, text=True, cwd=ROOT)
    assert one.returncode == 0 and grp.returncode == 0, one.stderr + grp.stderr
    a = re.search(r"closest=(\d+) shadow=(\d+) ms=\S+ checksum=(\S+)", one.stdout)
    b = re.search(r"group=2 frames=(\d+) completed=(\d+) closest=(\d+) shadow=(\d+) checksum=(\S+) reduce=\"(.*)\"", grp.stdout)
    assert a and b, one.stdout + grp.stdout
    assert int(b.group(1)) == frames and int(b.group(2)) == frames
    assert (a.group(1), a.group(2), a.group(3)) == (b.group(3), b.group(4), b.group(5))     # same rays, same image (the checksum is printed from the same doubles)
    assert "peer copies" in b.group(6)
