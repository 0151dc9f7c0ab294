"""Generates tests/golden/*.npz from the CPU oracle (the reference cannot run here and holds no
golden vectors of its own — SURVEY §4).  Re-run with `python tools/make_golden.py`; the fixtures
are small and committed.  Inputs that are not in the repo's assets are stored with the outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metal_raytracing_amd as m
import oracle as O

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)

# 1. Halton table slice: i in a spread of indices, d in 0..21
ii = np.array([0, 1, 2, 3, 5, 10, 1000, 123456, 1048575, 1048576 + 63], np.int32)
tab = np.array([[O.halton(int(i), d) for d in range(22)] for i in ii], np.float32)
np.savez_compressed(os.path.join(G, "halton.npz"), i=ii, table=tab)

# 2. DragonScene instance matrices (4x4 column-major) and default cameras
sc = m.DragonScene((1920, 1080))
mats = np.stack([O.make_transform(mo.position, mo.rotation, mo.scale) for mo in sc.models])
cams = {}
for (w, h) in [(1920, 1080), (256, 256), (800, 600)]:
    c = O.default_camera(w, h)
    cams[f"{w}x{h}"] = np.array([c.position.tolist(), c.right.tolist(), c.up.tolist(), c.forward.tolist()], np.float32)
np.savez_compressed(os.path.join(G, "dragonscene_setup.npz"), transforms=mats, **{"cam_" + k: v for k, v in cams.items()})

# 3. Cornell 64x64 spp1 and spp4 (3 bounces), plus per-bounce stage records of frame 0
csc = m.CornellScene((64, 64))
osc = O.OracleScene(m.flatten_scene(csc), csc.lights)
r = O.OracleRenderer(osc, 64, 64, seed=1, max_bounces=3, camera=csc.camera)
dump = r.render(1, dump=True)
a1 = r.accumulation().copy()
r.render(3)
a4 = r.accumulation().copy()
np.savez_compressed(os.path.join(G, "cornell64.npz"), spp1=a1, spp4=a4, stage=dump, counters=np.array(r.counters(), np.uint64))

# 4. seeds
seeds = np.array([O.seed_hash(1, i) for i in range(4096)], np.uint32)
np.savez_compressed(os.path.join(G, "seeds_seed1.npz"), seeds=seeds)

# 5. DragonScene 96x54 crop-size render with the procedural dragon replaced by nothing heavy:
#    train + treefir + planes + spheres only (the proxy mesh depends on libm sin/cos and is not pinned here)
class SmallDragonScene(m.Scene):
    def __init__(self, size):
        super().__init__(size)
        full = m.DragonScene(size)
        self.models = [mo for mo in full.models if mo.name != "dragon"]
ssc = SmallDragonScene((96, 54))
osc2 = O.OracleScene(m.flatten_scene(ssc), ssc.lights)
r2 = O.OracleRenderer(osc2, 96, 54, seed=1, max_bounces=3, camera=ssc.camera)
r2.render(2)
np.savez_compressed(os.path.join(G, "dragonscene_nodragon_96x54_spp2.npz"), accum=r2.accumulation(), counters=np.array(r2.counters(), np.uint64))
print("golden written:", sorted(os.listdir(G)))
