"""Ad-hoc GPU-vs-oracle comparison used during bring-up (the real gates are tests/ -m gpu)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metal_raytracing_amd as m
import oracle as O

def compare(name, scene_cls, w, h, frames=1, bounces=3, opts=None):
    sc = scene_cls((w, h))
    t = time.time(); r = m.Renderer((w, h), sc, max_bounces=bounces, scene_options=opts); t_build = time.time() - t
    st = r.device_scene.stats
    print(f"[{name}] tris={st.triangles} nodes={st.bvh_nodes} leaves={st.bvh_leaves} depth={st.max_depth} sah={st.sah_cost:.2f} build_ms={st.build_ms:.2f} (wall {t_build:.2f}s)", flush=True)
    r.draw(frames, wait=True)
    g = r.accumulation()
    rs = r.stats
    print(f"   gpu ms={rs.ms_gpu_last:.3f} closest={rs.closest_rays} shadow={rs.shadow_rays} ext_ms={rs.ms_extend_last:.3f}", flush=True)
    t = time.time(); os_ = O.OracleScene(m.flatten_scene(sc), sc.lights); orr = O.OracleRenderer(os_, w, h, max_bounces=bounces, camera=sc.camera)
    orr.render(frames); o = orr.accumulation(); print(f"   oracle {time.time()-t:.2f}s counters={orr.counters()}", flush=True)
    d = np.abs(g[..., :3] - o[..., :3])
    exact = np.all(g == o, axis=-1).mean()
    print(f"   bit-exact pixels {exact*100:.4f}%  maxabs {d.max():.3e}  rmse {np.sqrt((d**2).sum(-1).mean()):.3e}  within1e-3 {(d.max(-1) <= 1e-3).mean()*100:.4f}%", flush=True)
    r.close()
    return g, o

if __name__ == "__main__":
    which = sys.argv[1:] or ["cornell", "dragon_small"]
    if "cornell" in which:
        compare("cornell 256", m.CornellScene, 256, 256)
        compare("cornell 256 karras", m.CornellScene, 256, 256, opts={"builder": 0})
    if "dragon_small" in which:
        compare("dragon 480x270", m.DragonScene, 480, 270)
        compare("dragon 480x270 karras", m.DragonScene, 480, 270, opts={"builder": 0})
    if "dragon" in which:
        g, o = compare("dragon 1080p", m.DragonScene, 1920, 1080)
