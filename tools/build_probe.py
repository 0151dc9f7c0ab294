"""BVH build time of the benchmark scenes (device build, best of N; SceneStats.build_ms) and the tree it gives: tools/build_probe.py [--reps 5]"""
import argparse, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metal_raytracing_amd as mrt
import numpy as np

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=5); a = ap.parse_args()
ctx = mrt.Context(0)
S = mrt.SCENES
for name, scene, opts in (("cornell", S["cornell"]((256, 256)), {}), ("dragon", S["dragon"]((1920, 1080)), {}), ("dragon builder 0 (Karras)", S["dragon"]((1920, 1080)), {"builder": 0}),
                          ("dragon hostile", S["dragon_hostile"]((1920, 1080)), {}), ("garden", S["garden"]((3840, 2160)), {}),
                          ("dragon4 flat", S["dragon4"]((1920, 1080)), {}), ("dragon4 two-level", S["dragon4"]((1920, 1080)), {"instancing": 1})):
    best, st, wall, phases, recommit, moved = None, None, None, None, None, None
    import ctypes as C, time
    for _ in range(a.reps):
        d = mrt.DeviceScene(ctx, scene, opts)
        st = d.stats
        best = st.build_ms if best is None else min(best, st.build_ms)
        if wall is None or d.commit_wall_ms < wall:
            wall = d.commit_wall_ms
            ph = (C.c_double * 6)(); mrt._ffi.check(mrt.lib.mrt_debug_commit_times(d.handle, ph)); phases = list(ph)
        # a second commit of the SAME scene (an animated flattened scene rebuilds per frame): staging and scratch are already there
        mrt._ffi.check(mrt.lib.mrt_scene_set_option(d.handle, b"validate", 1.0))
        t0 = time.perf_counter(); d.commit(); dt = (time.perf_counter() - t0) * 1e3
        recommit = dt if recommit is None else min(recommit, dt)
        # a commit that only changes transforms (an animated scene): a flattened scene rebuilds from the geometry already on the device, a two-level scene rebuilds its TLAS
        xf = (C.c_float * 12)(); mrt._ffi.check(mrt.lib.mrt_scene_instance_transform(d.handle, 0, xf))
        m = np.eye(4, dtype=np.float32); m[:3, :] = np.array(list(xf), np.float32).reshape(4, 3).T; m = np.ascontiguousarray(m.T)
        d.set_instance_transform(0, m)
        t0 = time.perf_counter(); d.commit(); dt = (time.perf_counter() - t0) * 1e3
        moved = dt if moved is None else min(moved, dt)
        d.close()
    print(f"{name:26s} triangles {st.triangles:8d} build {best:7.3f} ms = {st.triangles / best / 1e3:6.1f} Mtris/s  commit wall {wall:7.2f} ms (staging {phases[0]:.2f} + allocs {phases[1]:.2f} + topology {phases[2]:.2f} + 8-wide {phases[3]:.2f} + rope {phases[4]:.2f} + validate {phases[5]:.2f}), re-commit {recommit:6.2f} ms, transform-only commit {moved:6.2f} ms  nodes {st.bvh_nodes} leaves {st.bvh_leaves} depth {st.max_depth} sah {st.sah_cost:.4f}", flush=True)
