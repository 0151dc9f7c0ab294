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


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["dragon", "garden"])
def test_write_through_refit_builds_the_tree_of_the_fenced_refit(mrt, gpu_ctx, scene_name):
    """The bottom-up pass of the build hands a node's box, cost and collapse table from one thread to another (possibly on another XCD, whose L2 is not coherent with this one's) through
    write-through stores, s_waitcnt and sc1 loads instead of __threadfence().  Scene option refit_fenced = 1 selects the fenced form: both must give the same 8-wide nodes — every
    quantised box, child mask and leaf range — bit for bit, on the 885 K-triangle scene (and a second one), twice."""
    import ctypes as C
    sc = mrt.SCENES[scene_name]((64, 64))
    def nodes(opts):
        d = mrt.DeviceScene(gpu_ctx, sc, opts)
        n = C.c_uint64()
        mrt._ffi.check(mrt.lib.mrt_debug_read_wnodes(d.handle, None, 0, C.byref(n)))
        out = np.zeros((n.value, 20), np.uint32)
        mrt._ffi.check(mrt.lib.mrt_debug_read_wnodes(d.handle, mrt._ffi.ptr(out), out.nbytes, C.byref(n)))
        st = d.stats
        d.close()
        return out, (st.bvh_nodes, st.bvh_leaves, st.wide_depth, st.sah_cost)
    fast, fast_stats = nodes({})
    for rep in range(2):
        fenced, fenced_stats = nodes({"refit_fenced": 1})
        assert fenced_stats == fast_stats
        assert fenced.shape == fast.shape
        # nodes of one level are numbered in the order their parents' waves reserve them (an atomic): a run-to-run permutation inside a level.  Compared as sets: everything
        # of a node but its child / packet base (words 4, 5) — origin, exponents, child mask, leaf ranges, all 48 quantised plane bytes
        def canon(a):
            rows = np.ascontiguousarray(a[:, [0, 1, 2, 3] + list(range(6, 20))])
            return rows[np.lexsort(rows.T[::-1])]
        assert np.array_equal(canon(fenced), canon(fast))
    assert fast.shape[0] > 1000
