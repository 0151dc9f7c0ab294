"""The device build at the sizes where its phases change hands: 1 .. 3 triangles, around the 1 024 clusters at which the last PLOC rounds move into one
workgroup (k_ploc_tail), around the batches of four rounds in front of it, and a size that needs several batches — every builder, flattened; closest-hit
and any-hit queries through both traversals against the oracle's brute force, bit for bit; the tree's own statistics must be those of a second build."""
import numpy as np
import pytest

from test_fuzz_geometry import _Raw, _rays


def _soup(rng, n):
    c = rng.uniform(-1, 1, (n, 3)) * [1.6, 0.9, 1.4] + [0, 1.0, 0]
    size = 10.0 ** rng.uniform(-2.2, -0.6, (n, 1, 1))
    tri = c[:, None, :] + rng.normal(size=(n, 3, 3)) * size
    return tri.reshape(-1, 3).astype(np.float32), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)


def _scene(mrt, n, seed):
    rng = np.random.default_rng(seed)
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [_Raw(mrt, "soup", *_soup(rng, n), (0.6, 0.6, 0.6), [0, 0, 0], [0, 0, 0], 1.0)]
    return S((64, 64))


@pytest.mark.gpu
@pytest.mark.parametrize("builder", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 3, 63, 65, 1023, 1024, 1025, 1026, 2047, 2049, 4097, 16385, 70001])
def test_build_at_phase_boundaries(mrt, orc, gpu_ctx, n, builder):
    sc = _scene(mrt, n, 7 * n + builder)
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    opts = {"builder": builder, "presplit": 0}
    d = mrt.DeviceScene(gpu_ctx, sc, opts)
    st = d.stats
    assert st.triangles == n and st.bvh_leaves >= 1
    rays = _rays(np.random.default_rng(n), 6000)
    o = osc.intersect_closest(rays)
    for g in (d.intersect_closest(rays), d.intersect_stream(rays)):
        for f in ("type", "distance", "primitive_id", "u", "v"):
            assert np.array_equal(g[f], o[f]), (f, n, builder)
    rays[:, 7] = 3.0
    assert np.array_equal(d.intersect_any(rays), osc.intersect_any(rays))
    d2 = mrt.DeviceScene(gpu_ctx, sc, opts)                  # node ids inside the build are handed out in arrival order: the emitted tree must not depend on it
    s2 = d2.stats
    assert (s2.bvh_nodes, s2.bvh_leaves, s2.max_depth, s2.sah_cost) == (st.bvh_nodes, st.bvh_leaves, st.max_depth, st.sah_cost)
    d.close(); d2.close()


@pytest.mark.gpu
def test_presplit_build_at_the_tail_boundary(mrt, orc, gpu_ctx):
    """references (not triangles) are the build's leaves: pre-splitting moves a 1 000-triangle mesh with slivers across the 1 024-cluster boundary"""
    rng = np.random.default_rng(5)
    pos, idx = _soup(rng, 1000)
    tri = pos.reshape(-1, 3, 3)
    tri[::9, 1] = tri[::9, 0] + [2.5, 0.0, 0.01]           # long slivers: split into several references each
    tri[::9, 2] = tri[::9, 0] + [0.0, 0.02, 0.0]
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [_Raw(mrt, "slivers", tri.reshape(-1, 3).astype(np.float32), idx, (0.5, 0.5, 0.5), [0, 0, 0], [0, 0, 0], 1.0)]
    sc = S((64, 64))
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    rays = _rays(np.random.default_rng(6), 6000)
    o = osc.intersect_closest(rays)
    for presplit in (0, 4):
        d = mrt.DeviceScene(gpu_ctx, sc, {"presplit": presplit})
        g = d.intersect_stream(rays)
        for f in ("type", "distance", "primitive_id", "u", "v"):
            assert np.array_equal(g[f], o[f]), (f, presplit)
        d.close()
