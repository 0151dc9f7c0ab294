"""The FIRST commit of a process (what bench.py's config.scene_commit_wall_ms reports), by phase: DragonScene as the only scene of a fresh process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import metal_raytracing_amd as mrt
def phases(d):
    ph = (C.c_double * 6)(); mrt._ffi.check(mrt.lib.mrt_debug_commit_times(d.handle, ph)); return list(ph)
sc = mrt.DragonScene((1920, 1080))
ctx = mrt.Context(0)
ds = mrt.DeviceScene(ctx, sc, {})
t = phases(ds)
print(f"first commit of the process: wall {ds.commit_wall_ms:.2f} ms = staging {t[0]:.2f} + allocs {t[1]:.2f} + topology {t[2]:.2f} + 8-wide {t[3]:.2f} + rope {t[4]:.2f} + validate {t[5]:.2f}; build (device) {ds.stats.build_ms:.2f} ms")
ds2 = mrt.DeviceScene(ctx, sc, {})
t = phases(ds2)
print(f"second scene, same process:  wall {ds2.commit_wall_ms:.2f} ms = staging {t[0]:.2f} + allocs {t[1]:.2f} + topology {t[2]:.2f} + 8-wide {t[3]:.2f} + rope {t[4]:.2f} + validate {t[5]:.2f}; build (device) {ds2.stats.build_ms:.2f} ms")
