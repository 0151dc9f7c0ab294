"""Round 6, VERDICT item 6: a chain of nested triangles (the scene that used to fall back to the rope kernels) — rate on the 8-wide layout it now keeps, on the rope layout
(scene option wide = 0: the old fallback), and of a scene of the same size whose triangles lie side by side (a balanced tree)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metal_raytracing_amd as mrt
from test_fuzz_geometry import _Raw
from test_deep_tree import _chain
w, h = 1280, 720
ctx = mrt.Context(0)
def grid(n):
    k = np.arange(n); gx, gz = k % 45, k // 45
    base = np.array([[0.0, 0.0, 0.0], [0.03, 0.0, 0.0], [0.0, 0.03, 0.0]])
    tri = base[None] + np.stack([-0.9 + gx * 0.04, np.full(n, 0.05), -0.9 + gz * 0.04], 1)[:, None, :]
    return tri.reshape(-1, 3).astype(np.float32), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
def scene(geom):
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [_Raw(mrt, "g", *geom, (0.7, 0.6, 0.5), [0, 0, 0], [0, 0, 0], 1.0), mrt.Model(name="plane", position=[0, 0, 0]), mrt.Model(name="sphere", position=[0.6, 0.4, 1.0], scale=0.4)]
    return S((w, h))
for n, growth in ((520, 1.0155), (2000, 1.004)):
    for label, geom, sopt in (("chain, 8-wide layout (radix-tree rebuild)", _chain(n, growth), None), ("chain, rope layout (the old fallback)", _chain(n, growth), {"wide": 0}), ("same count side by side, 8-wide layout", grid(n), None)):
        r = mrt.Renderer((w, h), scene(geom), ctx=ctx, scene_options=sopt)
        st = r.device_scene.stats
        r.draw(24, wait=True); r.reset_stats()
        t0 = time.perf_counter(); r.draw(96, wait=True); dt = time.perf_counter() - t0
        s = r.stats
        print(f"n = {n:5d}  {label:45s} wide_layout {st.wide_layout} depth {st.wide_depth:3d}  {(s.closest_rays + s.shadow_rays) / dt / 1e9:6.2f} Grays/s", flush=True)
        r.close()
