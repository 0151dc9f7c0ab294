"""The stream traversal kernels read other lanes' registers — the refill takes prefetched rays by ds_bpermute, the cooperative drain reads a straggler's ray by v_readlane — so a
register's value must survive in lanes that are switched off.  Hardware keeps it; a spill store or reload inside divergent control flow covers the active lanes only.  Two variant
builds of round 6 that pushed the persistent kernel from 24 to 60 / 64 bytes of scratch (7 waves per SIMD; the fp16-plane nodes at 6) rendered wrong images in the GPU suite
(profiles/r06_wide16_ab.txt).  The cause: the drain's helpers read their owner's ray with v_readlane INSIDE `if (helping)`, where the owner's lane is off — correct only as long as
the compiler never reloads that register there.  The reads now sit where the whole wave is on (traverse_wide.h), and the 7-wave variant passes the suite; every other cross-lane read
of the library was already at wave-uniform level or names an active lane.  This test compiles the device code (no GPU needed) and keeps an early warning beside the parity suite:
the kernels built on traverse_wide_stream spill at most 32 bytes — the shipped build's 24 are loop-invariant values stored at kernel entry — and never store a spill inside their loop."""
import os, re, shutil, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "metal-raytracing_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
STREAM_KERNELS = ("k_trace_mixed_wide_persist", "k_trace_mixed_wide_stream", "k_tl_top", "k_tl_blas", "k_query_stream", "k_trace_primary_wide_stream")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_stream_kernels_spill_only_at_entry(tmp_path):
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"): flags = [f for f in line.split("=", 1)[1].split() if not f.startswith("-W")]
    assert flags and "-ffp-contract=off" in flags
    s = str(tmp_path / "renderer.s")
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", *flags, "--cuda-device-only", "-S", "-o", s, os.path.join(CSRC, "renderer.hip")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    txt = open(s).read()
    seen = 0
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\s*\.amdhsa_kernel \1", txt, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        if not any(k in name for k in STREAM_KERNELS): continue
        seen += 1
        first_loop = next((i for i, l in enumerate(body) if "in Loop:" in l), len(body))
        stores = [i for i, l in enumerate(body) if re.search(r"\b(scratch_store|buffer_store)\w*\b.*Spill", l)]
        assert all(i < first_loop for i in stores), f"{name}: a register is spilled inside the traversal loop (line {stores[-1]} of the kernel, loop from {first_loop})"
        size = re.search(rf"\.set {re.escape(name)}\.private_seg_size, (\d+)", txt)
        assert size and int(size.group(1)) <= 32, f"{name}: {size.group(1) if size else '?'} bytes of scratch"
    assert seen >= 6, seen
