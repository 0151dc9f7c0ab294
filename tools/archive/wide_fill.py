"""Diagnostics: how full are the 8-wide nodes of a scene's BVH?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metal_raytracing_amd as m
from metal_raytracing_amd._ffi import lib, check, ptr
for name in sys.argv[1:] or ["dragon"]:
    sc = m.SCENES[name]((64, 64)); ctx = m.Context(0); ds = m.DeviceScene(ctx, sc)
    h = np.zeros(12, np.uint32); check(lib.mrt_debug_wide_histogram(ds.handle, ptr(h)))
    n = h[:9].sum()
    print(name, "wide nodes", n, "children/node", round((h[9] + h[10]) / n, 2), "inner", h[9], "leaf children", h[10], "tris", h[11], "tris/leaf child", round(h[11] / max(1, h[10]), 2),
          "histogram %", [round(100 * x / n, 1) for x in h[:9]])
