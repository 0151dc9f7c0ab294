"""Long accumulation against the oracle: hundreds of frames in uneven draw calls (passes of 4 on 12 streams, partial passes, the per-pixel
primary-ray hints carried from frame to frame, Halton indices well past the seed range) must still give the oracle's running average bit for bit."""
import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("scene,frames", [("dragon", 700), ("cornell", 1200)])
def test_long_accumulation_is_bit_identical(mrt, orc, gpu_ctx, scene, frames):
    w, h = 64, 36
    sc = mrt.SCENES[scene]((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, seed=7)
    done = 0
    for chunk in (1, 7, 100, 333):
        r.draw(chunk, wait=True); done += chunk
    r.draw(frames - done, wait=True)
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    ref = orc.OracleRenderer(osc, w, h, camera=sc.camera, seed=7); ref.render(frames)
    assert r.frameIndex == frames
    assert np.array_equal(r.accumulation(), ref.accumulation())
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    r.close()
