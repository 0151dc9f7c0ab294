"""Refit against build on DragonScene (885 K triangles): device time (MRTSceneStats.build_ms) and host wall time of the commit after mrt_scene_update_mesh, and the rate of a
48-frame draw on the refitted tree against a fresh build of the same deformed scene (a refit keeps the tree's shape: the larger the deformation, the looser its boxes)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
sc = mrt.DragonScene((w, h)); ctx = mrt.Context(0)
meshes = mrt.flatten_scene(sc, share=True)
k = max(range(len(meshes)), key=lambda i: len(meshes[i][0]))
pos0, nrm0 = np.asarray(meshes[k][0], np.float32), np.asarray(meshes[k][1], np.float32)
def deform(amp, phase):
    wv = (amp * np.sin(9.0 * pos0[:, 1] + phase) * np.cos(7.0 * pos0[:, 0] - phase)).astype(np.float32)
    return (pos0 + nrm0 * wv[:, None]).astype(np.float32), nrm0
def rate(r):
    best = 0
    for rep in range(3):
        r.draw(8, wait=True); r.reset_stats(); t0 = time.perf_counter(); r.draw(48, wait=True); dt = time.perf_counter() - t0
        st = r.stats; best = max(best, (st.closest_rays + st.shadow_rays) / dt / 1e6)
    return best
r = mrt.Renderer((w, h), sc, ctx=ctx); ds = r.device_scene
print(f"build: {ds.stats.build_ms:.2f} ms device, commit {ds.commit_wall_ms:.2f} ms wall; {rate(r):.0f} Mrays/s", flush=True)
for amp in (0.005, 0.02, 0.05, 0.1):
    p, n = deform(amp, 0.7)
    ds.update_mesh(k, p, n); t0 = time.perf_counter(); ds.commit(); wall = (time.perf_counter() - t0) * 1e3
    assert ds.refits >= 1
    rr = rate(r)
    f = mrt.Renderer((w, h), sc, ctx=ctx, scene_options={"refit": 0}); f.device_scene.update_mesh(k, p, n); t0 = time.perf_counter(); f.device_scene.commit(); fwall = (time.perf_counter() - t0) * 1e3
    fr = rate(f)
    cs = ds.stats
    print(f"amplitude {amp}: wide_cost {cs.wide_cost:.2f} / built {cs.wide_cost_built:.2f} = {cs.wide_cost / cs.wide_cost_built:.3f}, leaf_growth {cs.leaf_growth:.2f}; refit {ds.stats.build_ms:.2f} ms device, commit {wall:.2f} ms wall, {rr:.0f} Mrays/s   |   fresh build {f.device_scene.stats.build_ms:.2f} ms device, commit {fwall:.2f} ms wall, {fr:.0f} Mrays/s   (refit / build rate {rr / fr:.3f})", flush=True)
    f.close()
    ds.update_mesh(k, pos0, nrm0); ds.commit()          # back to the rest pose (a refit again): every row deforms the tree as built
    print(f"      back at the rest pose: leaf_growth {ds.stats.leaf_growth:.3f}, wide_cost ratio {ds.stats.wide_cost / ds.stats.wide_cost_built:.4f}", flush=True)

r.close()
# ---- two-level: dragon x 4 as ONE mesh + four instances — the BLAS refitted in place (both layouts) + the TLAS (round 6)
sc4 = mrt.InstancedDragonScene((w, h))
m4 = mrt.flatten_scene(sc4, share=True)
k4 = [i for i, m in enumerate(m4) if len(m[0]) > 100000 and m[4] < 0][0]
pos0, nrm0 = np.asarray(m4[k4][0], np.float32), np.asarray(m4[k4][1], np.float32)
r = mrt.Renderer((w, h), sc4, ctx=ctx, scene_options={"instancing": 1}); ds = r.device_scene
print(f"two-level build: {ds.stats.build_ms:.2f} ms device, commit {ds.commit_wall_ms:.2f} ms wall; {rate(r):.0f} Mrays/s", flush=True)
for amp in (0.005, 0.02, 0.05):
    p, n = deform(amp, 0.7)
    ds.update_mesh(k4, p, n); t0 = time.perf_counter(); ds.commit(); wall = (time.perf_counter() - t0) * 1e3
    assert ds.refits >= 1
    rr = rate(r); cs = ds.stats
    f = mrt.Renderer((w, h), sc4, ctx=ctx, scene_options={"instancing": 1, "refit": 0}); f.device_scene.update_mesh(k4, p, n); t0 = time.perf_counter(); f.device_scene.commit(); fwall = (time.perf_counter() - t0) * 1e3
    fr = rate(f)
    print(f"two-level amplitude {amp}: wide_cost ratio {cs.wide_cost / cs.wide_cost_built:.3f}, leaf_growth {cs.leaf_growth:.2f}; refit {cs.build_ms:.2f} ms device, commit {wall:.2f} ms wall, {rr:.0f} Mrays/s   |   fresh build {f.device_scene.stats.build_ms:.2f} ms device, commit {fwall:.2f} ms wall, {fr:.0f} Mrays/s   (refit / build rate {rr / fr:.3f})", flush=True)
    f.close()
    ds.update_mesh(k4, pos0, nrm0); ds.commit()
r.close()
