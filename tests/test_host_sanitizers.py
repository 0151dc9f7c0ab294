"""The host-side code that parses untrusted input or runs on threads, built with AddressSanitizer + UndefinedBehaviorSanitizer and driven on the CPU (GPU sanitizers are not
available on the pool): the OBJ / MTL reader (csrc/host_geometry.cpp, the stand-in for ModelIO's importer, Model.swift:16-21) over mutated copies of the shipped assets, and the
host binned-SAH builder (csrc/bvh_host_sah.cpp) over random and degenerate box sets — NaN, infinities, overflowing extents, identical boxes — with its topology invariants checked.
Round 6 found one defect this way: a NaN reaching a float -> int conversion in the builder's binning (undefined behaviour; benign on x86).
(The oracle built with the same two sanitizers runs its own CPU tests clean — tests/test_oracle_kat.py, test_independent_f64.py, test_fuzz_geometry.py, test_materials.py, test_instancing.py: 48 passed.)
(The whole library with its host code instrumented runs the CPU tests clean, but cannot run on the GPU box: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and
aborts at the first device allocation — the GPU sanitizers are not available on this pool.)"""
import os, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "metal-raytracing_amd", "csrc")
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + CSRC]
sys.path.insert(0, os.path.join(ROOT, "tests", "sanitize"))

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")


def _build(tmp_path, name, sources, extra=()):
    exe = str(tmp_path / name)
    p = subprocess.run(["g++", *SAN, *extra, "-o", exe, *sources, "-lpthread"], capture_output=True, text=True)
    if p.returncode != 0 and "sanitize" in p.stderr: pytest.skip("g++ without the sanitizer runtimes")
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


def test_obj_reader_under_asan_ubsan_on_mutated_assets(tmp_path):
    import obj_mutate
    exe = _build(tmp_path, "obj_harness", [os.path.join(ROOT, "tests", "sanitize", "obj_harness.cpp"), os.path.join(CSRC, "host_geometry.cpp")])
    shipped = sorted(os.path.join(obj_mutate.SRC, f) for f in os.listdir(obj_mutate.SRC) if f.endswith(".obj"))
    assert subprocess.run([exe] + shipped, capture_output=True, text=True).returncode == 0          # the assets themselves: loaded, indices in range, normals per vertex
    files = obj_mutate.make_cases(str(tmp_path / "cases"), seed=11, N=300)
    fails = obj_mutate.run_cases(exe, files)
    assert not fails, fails[0]


def test_host_sah_builder_under_asan_ubsan_on_degenerate_boxes(tmp_path):
    hip_inc = "/opt/rocm/include"
    if not os.path.isdir(hip_inc): pytest.skip("no HIP headers")
    exe = _build(tmp_path, "sah_harness", [os.path.join(ROOT, "tests", "sanitize", "sah_harness.cpp"), os.path.join(CSRC, "bvh_host_sah.cpp")], ["-D__HIP_PLATFORM_AMD__", "-I" + hip_inc])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
