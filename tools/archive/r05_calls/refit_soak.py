"""Repeated refits (an animation): 300 commits after mrt_scene_update_mesh with a travelling wave, then back to the rest pose — the rate on the tree must come back to what the
first refit of the rest pose gives (boxes of leaves that did not move must not creep), and the image must equal a fresh build's."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import metal_raytracing_amd as mrt
w, h = 1920, 1080
sc = mrt.DragonScene((w, h)); ctx = mrt.Context(0)
meshes = mrt.flatten_scene(sc, share=True)
k = max(range(len(meshes)), key=lambda i: len(meshes[i][0]))
pos0, nrm0 = np.asarray(meshes[k][0], np.float32), np.asarray(meshes[k][1], np.float32)
def deform(amp, phase):
    wv = (amp * np.sin(9.0 * pos0[:, 1] + phase) * np.cos(7.0 * pos0[:, 0] - phase)).astype(np.float32)
    return (pos0 + nrm0 * wv[:, None]).astype(np.float32)
def rate(r):
    best = 0
    for rep in range(3):
        r.draw(8, wait=True); r.reset_stats(); t0 = time.perf_counter(); r.draw(48, wait=True); dt = time.perf_counter() - t0
        st = r.stats; best = max(best, (st.closest_rays + st.shadow_rays) / dt / 1e6)
    return best
r = mrt.Renderer((w, h), sc, ctx=ctx); ds = r.device_scene
print(f"as built: {rate(r):.0f} Mrays/s", flush=True)
ds.update_mesh(k, pos0, nrm0); ds.commit(); print(f"rest pose after 1 refit: {rate(r):.0f} Mrays/s", flush=True)
t0 = time.perf_counter()
for f in range(300):
    ds.update_mesh(k, deform(0.01, 0.05 * f), nrm0); ds.commit()
print(f"300 refits: {(time.perf_counter() - t0) * 1e3 / 300:.2f} ms per update + commit (host deformation included); refits {ds.refits}", flush=True)
print(f"deformed (amplitude 0.01) after 300 refits: {rate(r):.0f} Mrays/s", flush=True)
ds.update_mesh(k, pos0, nrm0); ds.commit(); print(f"rest pose after 301 refits: {rate(r):.0f} Mrays/s", flush=True)
r.frameIndex = 0; r.draw(4, wait=True); a = r.accumulation().copy()
f = mrt.Renderer((w, h), sc, ctx=ctx); f.draw(4, wait=True)
print("image equals a fresh build's:", bool(np.array_equal(a.view(np.uint32), f.accumulation().view(np.uint32))), flush=True)
