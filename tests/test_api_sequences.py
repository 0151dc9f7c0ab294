"""Random sequences of the calls a host makes between frames — draw, deform a mesh, move an instance, new lights, new camera, resize, frame index back to 0, commit — run twice: once
with a wait after every step on a renderer with default options, once without waits on a renderer whose image-neutral options (pass size, lanes in flight, launch shapes, packed
shade, tile groups ...) are re-drawn at random between the steps.  Both must leave the same accumulation, bit for bit: whatever is still on the GPU when a call arrives, the call takes
effect behind it, and no option changes the image.  Flattened and two-level scenes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NEUTRAL = [("frame_batch", (1, 2, 3, 4, 8)), ("frames_in_flight", (1, 2, 3, 6)), ("persistent", (0, 1, 2)), ("persist_chunk", (64, 256, 1024)), ("wave_slots", (96, 1024, 0)), ("shade_pack", (0, 1)),
           ("tile_groups", (0, 1, 2, 3)), ("frame_bundle", (0, 1)), ("halton_table", (0, 1)), ("hit_lds", (0, 1)), ("stream_stride", (0, 1, 2)), ("fuse_primary", (0, 1, 2)), ("xcd_counters", (0, 1)), ("megakernel", (0, 0, 1)), ("primary_wide", (1, 2)), ("wide_bounce", (1, 1, 0))]


def _scene(mrt, size):
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [mrt.Model(name="plane", position=[0, 0, 0], scale=10), mrt.Model(name="sphere", position=[-1.2, 0.0, 0.4], scale=1),
                           mrt.Model(name="sphere", position=[1.4, 0.0, -0.2], rotation=[0.3, 1.1, -0.4], scale=1.7), mrt.Model(name="teapot", position=[0.1, 0.0, 1.6], rotation=[0, 0.6, 0], scale=0.01),
                           mrt.Model(name="teapot", position=[-0.6, 0.0, -1.0], rotation=[0, -1.3, 0], scale=0.013)]
    return S(size)


def _run(mrt, gpu_ctx, seed, two_level, fuzzed):
    from metal_raytracing_amd._ffi import _f3
    rng = np.random.default_rng(seed)              # the SAME stream of decisions in both runs
    orng = np.random.default_rng(1000 + seed)      # the fuzzed run's option changes
    size = (144, 81)
    sc = _scene(mrt, size)
    meshes = mrt.flatten_scene(sc, share=True)
    r = mrt.Renderer(size, sc, ctx=gpu_ctx, scene_options={"instancing": 1} if two_level else {})
    ds = r.device_scene
    own = [k for k, m in enumerate(meshes) if m[4] < 0 and len(m[0]) > 100]
    for step in range(14):
        if fuzzed:
            for _ in range(2):
                k, vals = NEUTRAL[orng.integers(len(NEUTRAL))]
                if two_level and k == "megakernel": continue          # (one launch per frame: flattened scenes only — a draw says so)
                try: r.set_option(k, float(vals[orng.integers(len(vals))]))
                except mrt.MRTError: pass          # (an option this scene kind does not take)
        op = rng.integers(8)
        if op <= 2: r.draw(int(rng.integers(1, 13)), wait=not fuzzed)
        elif op == 3:
            k = own[rng.integers(len(own))]; pos = np.asarray(meshes[k][0], np.float32); nrm = np.asarray(meshes[k][1], np.float32)
            amp = np.float32(rng.uniform(0.0, 0.04)); ph = np.float32(rng.uniform(0, 6))
            ds.update_mesh(k, (pos + nrm * (amp * np.sin(7.0 * pos[:, :1] + ph))).astype(np.float32), nrm); ds.commit()
        elif op == 4:
            k = int(rng.integers(1, len(meshes))); xf = np.array(meshes[k][2], np.float32).reshape(4, 4).copy(); xf[3, :3] += rng.uniform(-0.2, 0.2, 3).astype(np.float32)
            ds.set_instance_transform(k, xf); ds.commit()
        elif op == 5:
            lights = list(sc.lights); lights[0].color = _f3(rng.uniform(1, 8, 3)); ds.set_lights(lights)
            cam = sc.camera; cam.position = _f3((rng.uniform(-0.3, 0.3), rng.uniform(0.9, 1.2), 5.2)); r.set_camera(cam)
        elif op == 6: r.drawableSizeWillChange((int(rng.integers(40, 200)), int(rng.integers(30, 120))))
        else: r.frameIndex = int(rng.integers(0, 3))
    r.draw(3, wait=True)
    out = (r.accumulation().copy(), r.frameIndex)
    r.close()
    return out


@pytest.mark.parametrize("two_level", [False, True])
@pytest.mark.parametrize("seed", list(range(1, 9)))
def test_random_call_sequences_with_and_without_waits(mrt, gpu_ctx, seed, two_level):
    a, fa = _run(mrt, gpu_ctx, seed, two_level, fuzzed=False)
    b, fb = _run(mrt, gpu_ctx, seed, two_level, fuzzed=True)
    assert fa == fb and a.shape == b.shape
    assert np.isfinite(a).all() and float(a[..., :3].max()) > 0.0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
