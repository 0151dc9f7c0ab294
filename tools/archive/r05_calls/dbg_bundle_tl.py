import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import metal_raytracing_amd as mrt
w, h = 192, 108
ctx = mrt.Context(0)
for inst in (0, 1):
    sc = mrt.SCENES["dragon4"]((w, h))
    imgs = {}
    for fb, batch in ((0, 7), (1, 7), (1, 8), (1, 2)):
        r = mrt.Renderer((w, h), sc, ctx=ctx, scene_options={"instancing": inst})
        r.set_option("frame_bundle", fb); r.set_option("frame_batch", batch)
        r.draw(batch, wait=True); r.draw(2, wait=True) if batch == 7 else None
        if batch != 7: r.draw(9 - batch, wait=True)
        imgs[(fb, batch)] = r.accumulation().copy(); st = r.stats
        print("instancing", inst, "fb", fb, "batch", batch, "rays", st.closest_rays, st.shadow_rays, flush=True)
        r.close()
    ref = imgs[(0, 7)]
    for k, im in imgs.items():
        d = np.any(im != ref, axis=-1)
        ys, xs = np.nonzero(d)
        print("   ", k, "differing pixels", int(d.sum()), "first", list(zip(xs[:8].tolist(), ys[:8].tolist())), "max abs", float(np.abs(im - ref).max()))
