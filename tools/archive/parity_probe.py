"""Which build / option breaks bit parity?  One frame of DragonScene 320x180 (the bench contract's parity leg) against the oracle, per scene option set."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metal_raytracing_amd as mrt
import oracle as O
w, h = 320, 180
sc = mrt.DragonScene((w, h))
osc = O.OracleScene(mrt.flatten_scene(sc), sc.lights)
orr = O.OracleRenderer(osc, w, h, seed=1, max_bounces=3, camera=sc.camera); orr.render(1, threads=16)
ref = orr.accumulation()
ctx = mrt.Context(0)
for opts in [json.loads(a) for a in sys.argv[1:]] or [{}]:
    ropts = {k[2:]: v for k, v in opts.items() if k.startswith("r_")}
    sopts = {k: v for k, v in opts.items() if not k.startswith("r_")}
    r = mrt.Renderer((w, h), sc, ctx=ctx, seed=1, max_bounces=3, scene_options=sopts)
    for k, v in ropts.items(): r.set_option(k, v)
    r.draw(1, wait=True)
    g = r.accumulation(); st = r.stats
    bad = np.argwhere(~(g.view(np.uint32) == ref.view(np.uint32)).all(-1))
    print(os.environ.get("MRT_LIB_PATH", "default").split("_")[-1], opts, "differing pixels", len(bad), bad[:6].tolist(), "rays", (st.closest_rays, st.shadow_rays), "oracle", orr.counters(), flush=True)
    r.close()
