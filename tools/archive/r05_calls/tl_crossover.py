"""Two-level scenes: the tree-less TLAS pass (every lane visits every instance; <= 64 instances) against the walk of the 8-wide TLAS (tl_pairs = 2), by instance count."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import metal_raytracing_amd as mrt
from metal_raytracing_amd.scene import InstancedDragonScene
w, h = 1920, 1080
ctx = mrt.Context(0)
def rate(r):
    best = 0
    for rep in range(3):
        r.draw(8, wait=True); r.reset_stats(); t0 = time.perf_counter(); r.draw(48, wait=True); dt = time.perf_counter() - t0
        st = r.stats; best = max(best, (st.closest_rays + st.shadow_rays) / dt / 1e6)
    return best
for copies in (4, 8, 16, 24, 32, 48, 58):
    sc = InstancedDragonScene((w, h), copies=copies)
    out = []
    for tl in (1, 2):
        r = mrt.Renderer((w, h), sc, ctx=ctx, scene_options={"instancing": 1})
        r.set_option("tl_pairs", tl)
        out.append(rate(r)); n = r.device_scene.stats.instances
        r.close()
    print(f"{copies} dragons ({n} instances): flat TLAS pass {out[0]:.0f} Mrays/s, tree TLAS pass {out[1]:.0f} Mrays/s  (flat / tree {out[0] / out[1]:.3f})", flush=True)
