"""Deliberately deep trees: a chain of nested, growing triangles makes the agglomerative builder merge one pair per round — a binary tree as deep as the
scene has triangles, an 8-wide tree of about a seventh of that.  The render kernels size their LDS stack from the scene's depth; an agglomerative tree deeper than
48 wide levels is built again as a radix tree over the same Morton order (bounded depth whatever the geometry), so that EVERY such scene keeps the 8-wide layout
(round 6; until then a scene deeper than 64 levels fell back to the rope kernels at half the rate).  MRTSceneStats says what a scene got; the image is the oracle's."""
import numpy as np
import pytest

from test_fuzz_geometry import _Raw, _rays


def _chain(n, growth):
    """n right triangles that share their right-angle corner, each `growth` times the previous: the agglomerative builder can only ever merge the two smallest
    clusters (every other pair's box is the larger triangle's), so the binary tree is a chain n deep (simulated: 145 / 290 / 488 levels for n = 150 / 300 / 500)"""
    k = np.arange(n, dtype=np.float64)
    s = 1e-3 * growth ** k
    base = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    tri = base[None] * s[:, None, None] + np.array([-0.9, 0.05, -0.9])
    return tri.reshape(-1, 3).astype(np.float32), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)


def _scene(mrt, n, growth):
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [_Raw(mrt, "chain", *_chain(n, growth), (0.7, 0.6, 0.5), [0, 0, 0], [0, 0, 0], 1.0),
                           mrt.Model(name="plane", position=[0, 0, 0]), mrt.Model(name="sphere", position=[0.6, 0.4, 1.0], scale=0.4)]
    return S((96, 64))


@pytest.mark.gpu
@pytest.mark.parametrize("n,growth", [(150, 1.055), (300, 1.027), (520, 1.0155), (2000, 1.004)])
def test_deep_chain_renders_on_the_wide_layout(mrt, orc, gpu_ctx, n, growth):
    wide = True
    from test_gpu_parity import assert_parity, oracle_render
    w, h = 96, 64
    sc = _scene(mrt, n, growth)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    st = r.device_scene.stats
    print(n, "wide_layout", st.wide_layout, "wide_depth", st.wide_depth, "nodes", st.bvh_nodes)
    assert st.wide_layout == 1 and st.wide_depth <= 96, "every chain keeps the 8-wide layout"
    if n <= 300: assert st.wide_depth > 16, "the chain is meant to be deeper than the 16 levels the stack used to have (kept as the agglomerative builder made it: <= 48 levels)"
    rope = mrt.DeviceScene(gpu_ctx, sc, {"wide": 0})          # what such a scene used to fall back to: the same queries below must agree with it too
    r.draw(3, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 3)
    assert_parity(r.accumulation(), ref)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    # the query API and the render kernels' own walk on caller rays, against the oracle's brute force
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    rays = _rays(np.random.default_rng(n), 3000)
    o = osc.intersect_closest(rays, brute=True)
    walks = [r.device_scene.intersect_closest(rays), r.device_scene.intersect_stream(rays), rope.intersect_closest(rays)]
    for g in walks:
        for f in ("type", "distance", "primitive_id", "u", "v"):
            assert np.array_equal(g[f], o[f]), (f, n)
    if wide:                      # the low-latency mode walks the same layout with the same stack
        m = mrt.Renderer((w, h), sc, ctx=gpu_ctx); m.set_option("megakernel", 1); m.draw(3, wait=True)
        assert np.array_equal(m.accumulation(), r.accumulation())
        m.close()
    rope.close()
    r.close()
