"""SAH cost (device k_refit metric, root-normalised) of the committed tree per builder / leaf limit / scene."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metal_raytracing_amd as m
ctx = m.Context(0)
for name in ("dragon", "cornell"):
    sc = m.SCENES[name]((64, 64))
    for opts in ({"builder": 0}, {"builder": 1}, {"builder": 2}, {"builder": 1, "max_leaf": 1}, {"builder": 2, "max_leaf": 1}, {"builder": 1, "presplit": 0}, {"builder": 2, "presplit": 0}):
        ds = m.DeviceScene(ctx, sc, opts); s = ds.stats
        print(name, opts, "sah", round(s.sah_cost, 3), "nodes", s.bvh_nodes, "leaves", s.bvh_leaves, "depth", s.max_depth, "build ms", round(s.build_ms, 1), flush=True)
        ds.close()
