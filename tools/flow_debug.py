"""flow vs pipeline on a small frame: where do the images differ?"""
import sys, numpy as np
sys.path.insert(0, ".")
import importlib
mrt = importlib.import_module("metal_raytracing_amd")
W, H = 320, 200
sc = mrt.DragonScene((W, H))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opts = dict(kv.split("=") for kv in sys.argv[2:])
nbounce = int(opts.pop("bounces", 3))
a = mrt.Renderer((W, H), sc, max_bounces=nbounce); a.draw(frames, wait=True); ref = a.accumulation().copy(); sa = a.stats
b = mrt.Renderer((W, H), sc, ctx=a.ctx, max_bounces=nbounce); b.set_option("flow", 1)
for k, v in opts.items(): b.set_option(k, float(v))
b.draw(frames, wait=True); img = b.accumulation().copy()
try:
    sb = b.stats; print("rays", (sa.closest_rays, sa.shadow_rays), (sb.closest_rays, sb.shadow_rays))
except Exception as e:
    print("stats error:", e)
d = (img.view(np.uint32) != ref.view(np.uint32)).any(-1)
print("differing pixels", int(d.sum()), "of", d.size)
if d.any():
    ys, xs = np.nonzero(d); print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
    k = 0
    for y, x in list(zip(ys, xs))[:6]: print((y, x), img[y, x, :3], ref[y, x, :3])
    print("flow brighter:", int((img[..., :3].sum(-1) > ref[..., :3].sum(-1)).sum()), "darker:", int((img[..., :3].sum(-1) < ref[..., :3].sum(-1)).sum()))
