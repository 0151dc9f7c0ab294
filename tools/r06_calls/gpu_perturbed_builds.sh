#!/bin/bash
# round 6: the GPU suite against builds whose register allocation and inlining differ from the shipped one — what a latent dependence on a value surviving in a switched-off lane
# (tests/test_kernel_resources.py) or on an evaluation order would show up under.  Build the variants first (no GPU needed):
#   tools/build_variant.sh o1 "-O1"; tools/build_variant.sh o2ni "-O2 -mllvm -amdgpu-early-inline-all=false"
#   tools/build_variant.sh spill8 "-DMRT_SHADE_WAVES=8 -DMRT_SHADE_WIDE_WAVES=8 -DMRT_TWO_LEVEL_WAVES=8 -DMRT_WIDE_STREAM_WAVES=8"      (every kernel family at 64 registers: up to 92 B of scratch)
#   tools/build_variant.sh poison "-DMRT_POISON_ALLOC"      (scene_device.h: every device allocation and every piece of a build's scratch arena starts out as dwords of 1, not as zeros)
# Round 6, after the cooperative drain's cross-lane read was moved: 183 passed on each of the four (the three bench-contract tests, which time the library, left out).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; V=$R/metal-raytracing_amd/variants; O=gpurun_out/r06_perturbed; mkdir -p $O
for v in o1 o2ni spill8 poison; do [ -f $V/libmrt_hip_$v.so ] || continue
  MRT_LIB_PATH=$V/libmrt_hip_$v.so timeout -k 10 1100 python3 -m pytest tests -m gpu -q --deselect tests/test_bench_contract.py > $O/pytest_$v.log 2>&1; echo "$v:"; grep -n "^FAILED\|passed\|failed" $O/pytest_$v.log | head -30
done
