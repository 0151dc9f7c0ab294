"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  The bar: >= 99.5 % of pixels within 1e-3 abs/channel and
RGB-L2 RMSE <= 1e-3 at spp 1 (<= 2e-3 at spp 64).  Because both sides implement the same fixed
arithmetic contract the images are in fact expected to be BIT-IDENTICAL; the tests
assert the formal tolerance and, separately, a >= 99.99 % bit-exact fraction.

At the full BASELINE size the oracle is too slow for a per-test run, so the 1080p tests use
size-independent properties: builder invariance (two different BVHs give the same image), shard
additivity, ray-count conservation, accumulation identities, determinism."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_ABS, TOL_FRAC, TOL_RMSE = 1e-3, 0.995, 1e-3


def assert_parity(gpu, ref, rmse_tol=TOL_RMSE, exact_frac=0.9999):
    assert gpu.shape == ref.shape
    assert np.isfinite(gpu).all()
    d = np.abs(gpu[..., :3].astype(np.float64) - ref[..., :3])
    within = (d.max(-1) <= TOL_ABS).mean()
    rmse = np.sqrt((d ** 2).sum(-1).mean())
    assert within >= TOL_FRAC, f"only {within * 100:.3f}% of pixels within {TOL_ABS}"
    assert rmse <= rmse_tol, f"rmse {rmse}"
    exact = (gpu.view(np.uint32) == ref.view(np.uint32)).all(-1).mean()
    assert exact >= exact_frac, f"bit-exact fraction {exact}"
    assert np.all(gpu[..., 3] == ref[..., 3])


def oracle_render(orc, mrt, sc, w, h, frames, bounces=3, seed=1, shard=None, start_frame=0):
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    r = orc.OracleRenderer(osc, w, h, seed=seed, max_bounces=bounces, camera=sc.camera)
    if shard: r.set_shard(*shard)
    if start_frame: r.set_frame_index(start_frame)
    r.render(frames)
    return r.accumulation(), r.counters()


# ---------------------------------------------------------------- device helper functions
def test_device_halton_bit_exact(mrt, orc, gpu_ctx):
    import ctypes as C
    rng = np.random.default_rng(3)
    # the device takes digits two at a time for bases <= 23, one at a time up to i < 2^22, and runs the plain loop beyond
    i = np.concatenate([rng.integers(0, (1 << 20) + 4096, 20000), rng.integers(1 << 20, (1 << 22) + 64, 6000), [(1 << 22) - 2, (1 << 22) - 1, 1 << 22, (1 << 22) + 1, 528, 529, 530, 9 ** 6 - 1, 9 ** 6],
                        # non-positive indices (seed offset + frame index wrapped past 2^31): the reference's `while (i > 0)` gives 0
                        [-1, -2, -(1 << 20), -(1 << 31), 0, (1 << 31) - 1, (1 << 31) - 2, 1 << 30],
                        [0, 1, 2, 1048575, (1 << 24) - 1, 1 << 24, (1 << 24) + 12345]]).astype(np.int64).astype(np.int32)
    d = rng.integers(0, 22, len(i)).astype(np.int32); d[-8:] = [0, 1, 2, 16, 21, 99, 3, 50]; d[-16:-8] = [0, 1, 5, 21, 0, 0, 7, 2]
    out = np.zeros(len(i), np.float32)
    mrt._ffi.check(mrt.lib.mrt_debug_halton(gpu_ctx.handle, mrt._ffi.ptr(i), mrt._ffi.ptr(d), len(i), mrt._ffi.ptr(out)))
    ref = np.array([orc.halton(int(a), int(b)) for a, b in zip(i, d)], np.float32)
    assert np.array_equal(out, ref)
    g = np.load(os.path.join(GOLD, "halton.npz"))
    ii = np.repeat(g["i"], 22).astype(np.int32); dd = np.tile(np.arange(22, dtype=np.int32), len(g["i"]))
    out = np.zeros(len(ii), np.float32)
    mrt._ffi.check(mrt.lib.mrt_debug_halton(gpu_ctx.handle, mrt._ffi.ptr(ii), mrt._ffi.ptr(dd), len(ii), mrt._ffi.ptr(out)))
    assert np.array_equal(out.reshape(-1, 22), g["table"])


def test_device_hemisphere_and_seeds_bit_exact(mrt, orc, gpu_ctx):
    rng = np.random.default_rng(4)
    n = 5000
    u = rng.random((n, 2), dtype=np.float32); u[:4] = [[0, 1], [0, 0], [0.25, 0.5], [0.999999, 1e-8]]
    nr = rng.normal(size=(n, 3)).astype(np.float32); nr /= np.linalg.norm(nr, axis=1, keepdims=True); nr[0] = [0, 1, 0]
    out = np.zeros((n, 3), np.float32)
    mrt._ffi.check(mrt.lib.mrt_debug_hemisphere(gpu_ctx.handle, mrt._ffi.ptr(u), mrt._ffi.ptr(nr), n, mrt._ffi.ptr(out)))
    ref = np.stack([orc.align(orc.hemisphere(a, b), c) for (a, b), c in zip(u, nr)])
    assert np.array_equal(out, ref)
    seeds = np.zeros(64 * 64, np.uint32)
    mrt._ffi.check(mrt.lib.mrt_debug_seeds(gpu_ctx.handle, 1, 64, 64, mrt._ffi.ptr(seeds)))
    assert np.array_equal(seeds, np.load(os.path.join(GOLD, "seeds_seed1.npz"))["seeds"])
    mrt._ffi.check(mrt.lib.mrt_debug_seeds(gpu_ctx.handle, 77, 64, 64, mrt._ffi.ptr(seeds)))
    assert seeds[:50].tolist() == [orc.seed_hash(77, k) for k in range(50)] and seeds.max() < (1 << 20)


# ---------------------------------------------------------------- intersector (Raytracing.metal:244, :367)
def _rays(rng, n, lo, hi, tmax=np.inf):
    o = rng.uniform(lo - 1.0, hi + 1.0, (n, 3)).astype(np.float32)
    t = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32); rays[:, 0:3] = o; rays[:, 4:7] = d; rays[:, 7] = tmax
    return rays


@pytest.mark.parametrize("builder,wide", [(0, 0), (1, 0), (1, 1), (2, 1)])
@pytest.mark.parametrize("name,n", [("plane", 500), ("sphere", 3000), ("train", 3000), ("treefir", 2000), ("teapot", 1500)])
def test_intersect_matches_brute_force(mrt, orc, gpu_ctx, name, n, builder, wide):
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [mrt.Model(name=name, position=[0.1, -0.2, 0.3], rotation=[0.2, 0.5, -0.1], scale=1.3),
                           mrt.Model(name="plane", position=[0, -3, 0], scale=50)]
    sc = S((8, 8))
    ds = mrt.DeviceScene(gpu_ctx, sc, {"builder": builder, "wide": wide})
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    me = sc.meshes[0]
    w = (me.transform.T @ np.c_[me.positions, np.ones(len(me.positions))].T).T[:, :3]
    rays = _rays(np.random.default_rng(11), n, w.min(0), w.max(0))
    # axis-aligned and degenerate directions too
    rays[:6, 4:7] = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]]
    g, o = ds.intersect_closest(rays), osc.intersect_closest(rays, brute=True)
    for f in ("type", "instance_id", "geometry_id", "primitive_id"):
        assert np.array_equal(g[f], o[f]), f
    for f in ("distance", "u", "v"):
        assert np.array_equal(g[f].view(np.uint32), o[f].view(np.uint32)), f
    assert (g["type"] == 1).mean() > 0.3
    rays[:, 7] = np.random.default_rng(5).uniform(0.2, 6.0, len(rays)).astype(np.float32)
    rays[:, 3] = 0.05
    assert np.array_equal(ds.intersect_any(rays), osc.intersect_any(rays, brute=True))
    g, o = ds.intersect_closest(rays), osc.intersect_closest(rays, brute=True)
    assert np.array_equal(g["primitive_id"], o["primitive_id"]) and np.array_equal(g["distance"].view(np.uint32), o["distance"].view(np.uint32))
    ds.close()


def test_intersect_edge_cases(mrt, gpu_ctx):
    class Empty(mrt.Scene):
        pass
    sc = Empty((8, 8))                                              # no models at all
    ds = mrt.DeviceScene(gpu_ctx, sc)
    assert ds.stats.triangles == 0
    rays = _rays(np.random.default_rng(0), 100, np.zeros(3), np.ones(3))
    assert (ds.intersect_closest(rays)["type"] == 0).all() and (ds.intersect_any(rays) == 0).all()
    assert len(ds.intersect_closest(np.zeros((0, 8), np.float32))) == 0
    ds.close()

    class One(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [mrt.Model(name="plane", position=[0, 0, 0], scale=1), mrt.Model(name="plane", position=[0, 0, 0], scale=1)]
    sc = One((8, 8))
    ds = mrt.DeviceScene(gpu_ctx, sc)
    r = np.array([[0.3, 1, 0.2, 0, 0, -1, 0, np.inf], [0.3, -1, 0.2, 0, 0, 1, 0, np.inf], [5, 1, 0, 0, 0, -1, 0, np.inf], [0.3, 1, 0.2, 0, 0, -1, 0, 0.5]], np.float32)
    h = ds.intersect_closest(r)
    assert h["type"].tolist() == [1, 1, 0, 0] and h["instance_id"][:2].tolist() == [0, 0]      # tie → lowest id; no back-face culling
    assert h["distance"][0] == 1.0
    ds.close()


# ---------------------------------------------------------------- whole-frame parity
@pytest.mark.parametrize("builder", [0, 1, 2])
def test_cornell_256_spp1_parity(mrt, orc, gpu_ctx, builder):
    """BASELINE configs[0]."""
    sc = mrt.CornellScene((256, 256))
    r = mrt.Renderer((256, 256), sc, ctx=gpu_ctx, scene_options={"builder": builder})
    r.draw(1, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, 256, 256, 1)
    assert_parity(r.accumulation(), ref)
    st = r.stats
    assert (st.closest_rays, st.shadow_rays) == cnt and st.primary_rays == 256 * 256 and st.frames == 1
    r.close()


def test_cornell_golden_fixture(mrt, gpu_ctx):
    g = np.load(os.path.join(GOLD, "cornell64.npz"))
    sc = mrt.CornellScene((64, 64))
    r = mrt.Renderer((64, 64), sc, ctx=gpu_ctx)
    r.draw(1, wait=True)
    assert_parity(r.accumulation(), g["spp1"])
    r.draw(3, wait=True)
    assert_parity(r.accumulation(), g["spp4"])
    st = r.stats
    assert [st.closest_rays, st.shadow_rays] == g["counters"].tolist() and r.frameIndex == 4
    r.close()


def test_dragonscene_small_parity_spp1_and_accumulated(mrt, orc, gpu_ctx):
    """BASELINE configs[1] geometry (all 885 194 triangles) at a size the oracle finishes in seconds."""
    w, h = 320, 180
    sc = mrt.DragonScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    assert r.device_scene.stats.triangles == 885194 and r.device_scene.stats.instances == 7 and r.device_scene.stats.max_submeshes == 6
    r.draw(1, wait=True)
    ref1, cnt1 = oracle_render(orc, mrt, sc, w, h, 1)
    assert_parity(r.accumulation(), ref1)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt1
    r.draw(3, wait=True)
    ref4, cnt4 = oracle_render(orc, mrt, sc, w, h, 4)
    assert_parity(r.accumulation(), ref4)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt4
    r.close()


def test_dragonscene_4_bounces_parity(mrt, orc, gpu_ctx):
    """BASELINE configs[2] semantics (max_bounces = 4 → Halton dims to 21), reduced size/spp."""
    w, h = 192, 108
    sc = mrt.DragonScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=4)
    r.draw(8, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 8, bounces=4)
    assert_parity(r.accumulation(), ref, rmse_tol=2e-3)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    r.close()


def test_golden_dragonscene_without_dragon(mrt, gpu_ctx):
    g = np.load(os.path.join(GOLD, "dragonscene_nodragon_96x54_spp2.npz"))

    class SmallDragonScene(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [mo for mo in mrt.DragonScene(size).models if mo.name != "dragon"]
    r = mrt.Renderer((96, 54), SmallDragonScene((96, 54)), ctx=gpu_ctx)
    r.draw(2, wait=True)
    assert_parity(r.accumulation(), g["accum"])
    assert [r.stats.closest_rays, r.stats.shadow_rays] == g["counters"].tolist()
    r.close()


def test_all_light_types_parity(mrt, orc, gpu_ctx):
    """spot + sun + point + area (Raytracing.metal:281-327); configs[3] lights are spot + sun."""
    w, h = 160, 96
    sc = mrt.GardenScene((w, h))
    sc.lights = sc.lights + [mrt.Light.pointLight([0, 2.5, 1], [3, 2, 1]), mrt.Scene.setupLight()]
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    r.draw(2, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 2)
    assert_parity(r.accumulation(), ref)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt and ref[..., :3].max() > 0
    r.close()


def test_ragged_size_and_resize(mrt, orc, gpu_ctx):
    """Sizes that are not multiples of the 8x8 tile (the bounds check of Raytracing.metal:171) and
    drawableSizeWillChange (Renderer.swift:353-356): new seeds, frameIndex back to 0."""
    sc = mrt.CornellScene((37, 21))
    r = mrt.Renderer((37, 21), sc, ctx=gpu_ctx)
    r.draw(2, wait=True)
    ref, _ = oracle_render(orc, mrt, sc, 37, 21, 2)
    assert_parity(r.accumulation(), ref, exact_frac=1.0)
    r.drawableSizeWillChange((50, 19))
    assert r.frameIndex == 0
    r.draw(1, wait=True)
    sc2 = mrt.CornellScene((50, 19))
    ref2, _ = oracle_render(orc, mrt, sc2, 50, 19, 1)
    assert_parity(r.accumulation(), ref2, exact_frac=1.0)
    r.close()


def test_shards_sum_to_full_frame(mrt, orc, gpu_ctx):
    w, h = 200, 120
    sc = mrt.CornellScene((w, h))
    full = mrt.Renderer((w, h), sc, ctx=gpu_ctx); full.draw(2, wait=True); f = full.accumulation(); full.close()
    acc = np.zeros_like(f); rays = 0
    for rank in range(3):
        r = mrt.Renderer((w, h), sc, ctx=gpu_ctx); r.set_shard(rank, 3); r.draw(2, wait=True)
        a = r.accumulation()
        oa, _ = oracle_render(orc, mrt, sc, w, h, 2, shard=(rank, 3))
        assert np.array_equal(a, oa)
        acc += a; rays += r.stats.primary_rays; r.close()
    assert np.array_equal(acc, f) and rays == 2 * w * h


def test_shards_in_tile_groups_sum_to_full_frame(mrt, orc, gpu_ctx):
    """A sharded renderer whose passes carry one frame runs them as tile groups (group g of G of rank r of N = shard g N + r of G N): every rank's image is its shard of the
    oracle's, the shards sum to the full frame and the ray counts add up — also when a rank owns so few tiles that groups are refused (tiles_local < 64)."""
    for (w, h), world in (((512, 288), 3), ((96, 64), 3)):
        sc = mrt.CornellScene((w, h))
        full = mrt.Renderer((w, h), sc, ctx=gpu_ctx); full.draw(3, wait=True); f = full.accumulation(); full.close()
        acc = np.zeros_like(f); rays = 0
        for rank in range(world):
            r = mrt.Renderer((w, h), sc, ctx=gpu_ctx); r.set_shard(rank, world); r.set_option("frame_batch", 1); r.set_option("frames_in_flight", 1)
            r.draw(1, wait=True); r.draw(2, wait=True)
            assert r.get_option("groups_used") == (3 if w == 512 else 1)
            a = r.accumulation()
            oa, _ = oracle_render(orc, mrt, sc, w, h, 3, shard=(rank, world))
            assert np.array_equal(a, oa)
            acc += a; rays += r.stats.primary_rays; r.close()
        assert np.array_equal(acc, f) and rays == 3 * w * h


def test_tonemap_matches_oracle(mrt, orc, gpu_ctx):
    sc = mrt.CornellScene((64, 48))
    r = mrt.Renderer((64, 48), sc, ctx=gpu_ctx); r.draw(3, wait=True)
    assert np.array_equal(r.tonemapped(), orc.tonemap_rgba8(r.accumulation()))
    r.close()


def test_error_paths(mrt, gpu_ctx):
    import ctypes as C
    h = C.c_void_p()
    assert mrt.lib.mrt_renderer_create(gpu_ctx.handle, None, 8, 8, 1, 3, C.byref(h)) == 1
    assert b"bad argument" in mrt.lib.mrt_last_error()
    sc = mrt.CornellScene((16, 16))
    with pytest.raises(mrt.MRTError):
        mrt.Renderer((16, 16), sc, ctx=gpu_ctx, max_bounces=0)
    with pytest.raises(mrt.MRTError):
        mrt.Renderer((0, 16), sc, ctx=gpu_ctx)
    r = mrt.Renderer((16, 16), sc, ctx=gpu_ctx)
    buf = np.zeros(10, np.float32)
    assert mrt.lib.mrt_renderer_read_accum(r.handle, mrt._ffi.ptr(buf), buf.nbytes) == 1
    with pytest.raises(mrt.MRTError):
        r.set_shard(3, 2)
    sc.lights = []
    r2 = mrt.Renderer((16, 16), sc, ctx=gpu_ctx)
    with pytest.raises(mrt.MRTError) as e:
        r2.draw(1)
    assert e.value.code == 5
    r.close(); r2.close()
    with pytest.raises(mrt.MRTError):
        mrt.Context(999)
    # handles are destroyed inside out: a scene with a live renderer and a context with a live scene refuse (nothing is freed), and accept once their dependants are gone
    c2 = C.c_void_p(); assert mrt.lib.mrt_context_create(0, C.byref(c2)) == 0
    s2 = C.c_void_p(); assert mrt.lib.mrt_scene_create(c2, C.byref(s2)) == 0
    assert mrt.lib.mrt_context_destroy(c2) == 5 and b"still alive" in mrt.lib.mrt_last_error()
    assert mrt.lib.mrt_scene_commit(s2) == 0          # (an empty scene commits)
    r3 = C.c_void_p(); assert mrt.lib.mrt_renderer_create(c2, s2, 8, 8, 1, 3, C.byref(r3)) == 0
    assert mrt.lib.mrt_scene_destroy(s2) == 5 and b"still use this scene" in mrt.lib.mrt_last_error()
    assert mrt.lib.mrt_renderer_destroy(r3) == 0 and mrt.lib.mrt_context_destroy(c2) == 5
    assert mrt.lib.mrt_scene_destroy(s2) == 0 and mrt.lib.mrt_context_destroy(c2) == 0
    # a NaN or an infinity in a position, a normal or a transform is refused by the call that brings it (and leaves what was there)
    lib = mrt.lib; ptr = mrt._ffi.ptr
    s = C.c_void_p(); assert lib.mrt_scene_create(gpu_ctx.handle, C.byref(s)) == 0
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32); nrm = np.tile(np.array([0, 0, 1], np.float32), (3, 1)); xf = np.eye(4, dtype=np.float32); mid = C.c_int32()
    for bad in (np.nan, np.inf, -np.inf):
        p2 = pos.copy(); p2[1, 2] = bad; n2 = nrm.copy(); n2[2, 0] = bad; x2 = xf.copy(); x2[3, 1] = bad
        assert lib.mrt_scene_add_mesh(s, ptr(p2), 12, ptr(nrm), 12, 3, ptr(xf), C.byref(mid)) == 1 and b"NaN or infinite" in lib.mrt_last_error()
        assert lib.mrt_scene_add_mesh(s, ptr(pos), 12, ptr(n2), 12, 3, ptr(xf), C.byref(mid)) == 1
        assert lib.mrt_scene_add_mesh(s, ptr(pos), 12, ptr(nrm), 12, 3, ptr(x2), C.byref(mid)) == 1 and b"transform" in lib.mrt_last_error()
    assert lib.mrt_scene_add_mesh(s, ptr(pos), 12, ptr(nrm), 12, 3, ptr(xf), C.byref(mid)) == 0 and mid.value == 0
    big = pos * np.float32(3e38)          # finite, however large: accepted
    assert lib.mrt_scene_update_mesh(s, 0, ptr(big), 12, ptr(nrm), 12, 3) == 0 and lib.mrt_scene_update_mesh(s, 0, ptr(pos), 12, ptr(nrm), 12, 3) == 0
    p2 = pos.copy(); p2[0, 0] = np.nan; x2 = xf.copy(); x2[0, 0] = np.inf
    assert lib.mrt_scene_update_mesh(s, 0, ptr(p2), 12, ptr(nrm), 12, 3) == 1 and b"keeps what it had" in lib.mrt_last_error()
    assert lib.mrt_scene_set_instance_transform(s, 0, ptr(x2)) == 1 and lib.mrt_scene_add_instance(s, 0, ptr(x2), C.byref(mid)) == 1
    lib.mrt_scene_destroy(s)


# ---------------------------------------------------------------- BASELINE full size: properties
@pytest.fixture(scope="module")
def dragon1080(mrt, gpu_ctx):
    sc = mrt.DragonScene((1920, 1080))
    r = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx)
    r.draw(1, wait=True)
    yield sc, r, r.accumulation().copy()
    r.close()


def test_1080p_builder_invariance(mrt, gpu_ctx, dragon1080):
    """Two different trees (Karras radix tree vs PLOC) must give the same image bit for bit: the
    closest hit is the global minimum with a fixed tie-break, independent of traversal order."""
    sc, r, img = dragon1080
    r2 = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx, scene_options={"builder": 0, "max_leaf": 2})
    r2.draw(1, wait=True)
    assert np.array_equal(r2.accumulation(), img)
    assert (r2.stats.closest_rays, r2.stats.shadow_rays) == (r.stats.closest_rays, r.stats.shadow_rays)
    r2.close()
    # a third tree: binned SAH from the host builder, greedy 8-wide collapse, slivers pre-split into references
    r3 = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx, scene_options={"builder": 2, "wide_collapse": 0, "presplit": 2})
    r3.draw(1, wait=True)
    assert np.array_equal(r3.accumulation(), img)
    assert (r3.stats.closest_rays, r3.stats.shadow_rays) == (r.stats.closest_rays, r.stats.shadow_rays)
    r3.close()


def test_1080p_counts_and_determinism(mrt, gpu_ctx, dragon1080):
    sc, r, img = dragon1080
    st = r.stats
    npx = 1920 * 1080
    assert st.primary_rays == npx and npx <= st.closest_rays <= 3 * npx and 0 < st.shadow_rays <= st.closest_rays
    assert np.isfinite(img).all() and (img[..., :3] >= 0).all() and (img[..., 3] == 1).all()
    lit = (img[..., :3].sum(-1) > 0).mean()
    assert 0.2 < lit < 1.0
    r2 = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx); r2.draw(1, wait=True)
    assert np.array_equal(r2.accumulation(), img)                       # same seed → same image
    r3 = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx, seed=2); r3.draw(1, wait=True)
    assert not np.array_equal(r3.accumulation(), img)
    r2.close(); r3.close()


def test_1080p_crop_against_oracle(mrt, orc, gpu_ctx, dragon1080):
    """Oracle on a sparse subset of 8x8 tiles of the full-size frame (shard 0 of 97)."""
    sc, r, img = dragon1080
    ref, _ = oracle_render(orc, mrt, sc, 1920, 1080, 1, shard=(0, 97))
    tiles_x = 240
    ys, xs = np.mgrid[0:1080, 0:1920]
    own = (((ys // 8) * tiles_x + xs // 8) % 97) == 0
    assert own.sum() > 20000
    assert_parity(img[own][None], ref[own][None])


def test_1080p_accumulation_identity(mrt, gpu_ctx, dragon1080):
    """Frame k of an accumulation equals (sum of the k single frames)/k up to fp32 re-association:
    render frames 0..3 separately (frame index set explicitly), average, compare with spp 4."""
    sc, r, img = dragon1080
    ra = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx); ra.draw(4, wait=True); acc = ra.accumulation(); ra.close()
    singles = []
    for k in range(4):
        rk = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx)
        if k:
            rk.frameIndex = k
            z = np.zeros((1080, 1920, 4), np.float32)                    # prev = 0 → out = sample/(k+1)
        rk.draw(1, wait=True)
        singles.append(rk.accumulation()[..., :3].astype(np.float64) * (k + 1))
        rk.close()
    mean = sum(singles) / 4
    assert np.allclose(acc[..., :3], mean, rtol=1e-5, atol=1e-6)
    assert np.array_equal(singles[0].astype(np.float32), img[..., :3])


# ---------------------------------------------------------------- alternative traversal backends
@pytest.mark.parametrize("backend", ["default", "rope_only", "wide_primary_stream", "rope_primary_in_shade", "rope_bounce", "one_frame_in_flight", "eight_frames_in_flight", "one_frame_per_pass", "three_frames_per_pass", "eight_frames_per_pass",
                                     "no_primary_hint", "persistent_always", "persistent_never", "small_persistent_grid", "one_work_counter",
                                     "no_hit_lds", "no_hit_lds_static_split", "hit_lds_small_grid", "unpacked_shade", "packed_shade_one_frame_passes", "no_frame_bundle", "no_halton_table", "frame_bundle_passes_of_three", "frame_bundle_no_hint",
                                     "stream_stride_static_split"])
def test_traversal_backends_agree_with_oracle(mrt, orc, gpu_ctx, backend):
    """Every path a scene or option can reach must give the oracle's image: the default (every ray on the 8-wide layout; primary rays traced inside
    shade(0)), a scene without the 8-wide layout (rope kernels for everything), the primary rays on the 8-wide stream kernel / on the rope layout
    inside shade(0), bounce and shadow rays on the rope kernels (scene option rope = 1), and any pass shape / number of passes in flight."""
    w, h = 256, 144
    sc = mrt.DragonScene((w, h))
    sopt = {"wide": 0} if backend.startswith("rope_only") else {"rope": 1} if backend in ("rope_bounce", "rope_primary_in_shade") else None
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options=sopt)
    assert r.device_scene.stats.wide_layout == (0 if backend.startswith("rope_only") else 1)
    if backend == "wide_primary_stream": r.set_option("primary_wide", 1)
    if backend == "rope_primary_in_shade": r.set_option("primary_wide", 0)
    if backend == "rope_bounce": r.set_option("wide_bounce", 0)
    if backend == "no_primary_hint": r.set_option("primary_hint", 0)
    if backend == "persistent_always": r.set_option("persistent", 1); r.set_option("persist_chunk", 64)
    if backend == "persistent_never": r.set_option("persistent", 0)
    if backend == "one_work_counter": r.set_option("persistent", 1); r.set_option("persist_chunk", 64); r.set_option("xcd_counters", 0)      # (default: one counter and one eighth of every sub-frame's rays per XCD)
    if backend == "small_persistent_grid": r.set_option("persistent", 1); r.set_option("wave_slots", 96)      # a long drain phase on few waves
    # the pulling launch reports a finished ray from LDS (traverse_wide.h StreamExt, the default)
    if backend == "no_hit_lds": r.set_option("persistent", 1); r.set_option("persist_chunk", 64); r.set_option("hit_lds", 0)      # the pulling launch without LDS extras (round 4's kernel)
    if backend.startswith("packed_shade"):                  # k_shade of bounces 1, 2 compacts the hits of its queue in LDS and shades them on full waves (k_shade_pack)
        r.set_option("shade_pack", 1)          # (the default)
        if backend.endswith("one_frame_passes"): r.set_option("frame_batch", 1); r.set_option("frames_in_flight", 2)
    if backend == "no_halton_table": r.set_option("halton_table", 0)          # bounce 0's Halton values by the recurrence (default: a bundled pass reads them from the renderer's table, FrameParams::halton_tab)
    if backend == "no_frame_bundle": r.set_option("frame_bundle", 0)          # shade(0): a wave = the 64 pixels of one tile in one sub-frame (round 4's form)
    if backend.startswith("frame_bundle"):                  # shade(0) of a multi-frame pass: a wave takes 8 slots x 8 sub-frames (FrameParams::frame_bundle); 5 frames: three lanes of every eight idle
        r.set_option("frame_bundle", 1)          # (the default)
        if backend.endswith("passes_of_three"): r.set_option("frame_batch", 3)
        if backend.endswith("no_hint"): r.set_option("primary_hint", 0)
    if backend == "stream_stride_static_split": r.set_option("persistent", 0); r.set_option("stream_stride", 1)      # the static split deals 64-ray batches round-robin to the waves (traverse_wide.h BatchStride)
    if backend == "unpacked_shade": r.set_option("shade_pack", 0)          # one queue entry per thread, hit or miss (round 4's form)
    if backend == "no_hit_lds_static_split": r.set_option("persistent", 0); r.set_option("hit_lds", 0)              # the static split without them (persistent_never runs it with them: the default)
    if backend == "hit_lds_small_grid": r.set_option("persistent", 1); r.set_option("persist_chunk", 64); r.set_option("wave_slots", 96)      # few waves: a long drain phase, where idle lanes help a straggler and hand their hit to its owner
    if backend == "one_frame_in_flight": r.set_option("frames_in_flight", 1)
    if backend == "eight_frames_in_flight": r.set_option("frames_in_flight", 8)
    if backend.endswith("_per_pass"): r.set_option("frame_batch", {"one": 1, "three": 3, "eight": 8}[backend.split("_")[0]])   # default 8: 5 frames = one pass; three: 3 + 2
    r.draw(5, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 5)
    assert_parity(r.accumulation(), ref)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    if backend in ("hybrid_default", "rope_only"):          # the query entry points use the wide layout when the scene has one
        rays = _rays(np.random.default_rng(3), 4000, np.array([-2, 0, -1.5]), np.array([3, 2, 3]))
        osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
        g, o = r.device_scene.intersect_closest(rays), osc.intersect_closest(rays)
        assert np.array_equal(g["primitive_id"], o["primitive_id"]) and np.array_equal(g["distance"].view(np.uint32), o["distance"].view(np.uint32))
        rays[:, 7] = 1.5
        assert np.array_equal(r.device_scene.intersect_any(rays), osc.intersect_any(rays))
    r.close()


@pytest.mark.parametrize("groups,size", [(2, (256, 144)), (3, (333, 187)), (4, (256, 144)), (0, (256, 144)), (4, (200, 64))])
def test_tile_groups_render_the_same_image(mrt, orc, gpu_ctx, groups, size):
    """A short draw runs every pass as G groups of tiles on G lanes (renderer option tile_groups; renderer.h TileGroup): disjoint pixels, one accumulation target.  One frame
    alone, a draw of several one-frame passes, a batched pass, a draw after the group count changed and a long draw that falls back to whole passes must give the oracle's
    image and ray counts whatever G — also at ragged sizes and when a group gets few tiles."""
    w, h = size
    sc = mrt.DragonScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    r.set_option("tile_groups", groups)
    r.set_option("frame_batch", 1); r.set_option("frames_in_flight", 1)
    r.draw(1, wait=True)                                             # one frame alone: G groups on G lanes
    if groups >= 2 and w * h >= 256 * 144: assert r.get_option("groups_used") == groups
    if groups == 0: assert r.get_option("groups_used") == 3          # by the draw: a one-frame pass alone on the chip runs as three groups
    r.set_option("frames_in_flight", 2); r.draw(3, wait=True)        # passes of one frame, two in flight, each in groups
    r.set_option("frame_batch", 4); r.set_option("frames_in_flight", 6); r.draw(6, wait=True)      # 3 + 3 frames
    r.set_option("tile_groups", 2 if groups != 2 else 3); r.draw(2, wait=True)                     # another group count: other tiles per lane, new seed tables
    r.set_option("tile_groups", groups); r.draw(1, wait=True)
    r.set_option("tile_groups", 0); r.set_option("frame_batch", 1); r.draw(13, wait=True)          # more passes than lanes: whole passes (G = 1)
    assert r.get_option("groups_used") == 1
    assert r.frameIndex == 26
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 26)
    assert_parity(r.accumulation(), ref)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    r.close()


@pytest.mark.parametrize("scene_name,size", [("dragon", (333, 187)), ("cornell", (64, 64)), ("cornell", (1000, 3)), ("garden", (257, 129)), ("dragon_hostile", (320, 180))])
def test_packed_shade_at_ragged_sizes_and_other_scenes(mrt, orc, gpu_ctx, scene_name, size):
    """k_shade_pack (renderer option shade_pack) where the queue's length is no multiple of anything, on all four light types and on the hostile stand-in."""
    w, h = size
    sc = mrt.SCENES[scene_name]((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    r.set_option("shade_pack", 1)
    r.draw(3, wait=True); r.draw(2, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 5)
    assert_parity(r.accumulation(), ref)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    r.close()


@pytest.mark.parametrize("scene_name,size,batch", [("dragon", (333, 187), 8), ("cornell", (64, 64), 12), ("cornell", (1000, 3), 5), ("garden", (257, 129), 8), ("dragon_hostile", (320, 180), 6), ("dragon4", (192, 108), 7)])
def test_frame_bundle_at_ragged_sizes_batches_and_other_scenes(mrt, orc, gpu_ctx, scene_name, size, batch):
    """shade(0) with eight sub-frames of a slot side by side in a wave (renderer option frame_bundle): passes of more than eight frames (two groups per slot), of fewer
    (idle lanes), partial tiles, the image's centre row and column (bundles whose directions change sign walk one ray per lane), the hostile stand-in and a two-level scene
    (dragon x 4 as instances) must give the oracle's image and counts."""
    w, h = size
    sc = mrt.SCENES[scene_name]((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"instancing": 1} if scene_name == "dragon4" else None)
    r.set_option("frame_bundle", 1); r.set_option("frame_batch", batch)
    n = batch + 2
    r.draw(batch, wait=True); r.draw(2, wait=True)
    if scene_name == "dragon4":          # a two-level scene's rays are tested in object space: its oracle is the two-level one
        o = orc.OracleRenderer(orc.OracleScene(mrt.flatten_scene(sc, share=True), sc.lights, instancing=True), w, h, seed=1, max_bounces=3, camera=sc.camera)
        o.render(n); ref, cnt = o.accumulation(), o.counters()
    else: ref, cnt = oracle_render(orc, mrt, sc, w, h, n)
    assert_parity(r.accumulation(), ref)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    r.close()


def test_halton_table_window_moves_with_the_frame_index(mrt, gpu_ctx):
    """A bundled pass reads bounce 0's Halton values from a table over a window of indices (renderer.hip FrameParams::halton_tab, filled by the recurrence itself).  When the frame
    index leaves the window — 65 536 frames on, or back after a reset — the table is refilled: the image and the ray counts must be those of the recurrence throughout."""
    w, h = 160, 96
    sc = mrt.DragonScene((w, h))
    out = []
    for tab in (1, 0):
        r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
        r.set_option("halton_table", tab); r.set_option("frame_batch", 8)
        r.draw(8, wait=True); imgs = [r.accumulation().copy()]
        r.frameIndex = 65532; r.draw(8, wait=True); imgs.append(r.accumulation().copy())          # 65532 + 8 > 65536: the window moves
        r.frameIndex = 3; r.draw(5, wait=True); imgs.append(r.accumulation().copy())               # and back
        r.frameIndex = (1 << 31) - 4; r.draw(8, wait=True); imgs.append(r.accumulation().copy())   # indices wrap past 2^31 (halton = 0 there): outside any window
        out.append((imgs, (r.stats.closest_rays, r.stats.shadow_rays)))
        r.close()
    assert out[0][1] == out[1][1]
    for a, b in zip(out[0][0], out[1][0]): assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_frame_batch_can_change_between_draws(mrt, orc, gpu_ctx):
    """Frames travel through the pipeline in batches (frame_batch per pass); the running average must not depend on how
    the frames were grouped, nor on the option changing between draws (the buffers are re-sized, the image is kept)."""
    w, h = 200, 120
    sc = mrt.CornellScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    assert r.get_option("frame_batch") == 32   # the default goes by image size (8 at 1920 x 1080 and above, up to 32 below); reads back as the value in force
    r.draw(3, wait=True)                       # one pass of 3
    r.set_option("frame_batch", 2); r.draw(5, wait=True)    # 2 + 2 + 1
    r.set_option("frame_batch", 8); r.draw(3, wait=True)    # one pass of 3
    assert r.get_option("frame_batch") == 8
    r.set_option("frame_batch", 0); r.draw(40, wait=True)   # back to the default: 32 here, and a draw's passes take at most a third of it each (14 + 13 + 13)
    assert r.get_option("frame_batch") == 32
    assert r.frameIndex == 51
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 51)
    assert_parity(r.accumulation(), ref)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    with pytest.raises(mrt.MRTError):
        r.set_option("frame_batch", 33)
    r.close()


@pytest.mark.parametrize("size", [(333, 187), (64, 64), (1000, 3), (257, 129)])
def test_chunk_pulling_covers_every_ray_whatever_the_rounding(mrt, gpu_ctx, size):
    """The pulling traversal launches cut the combined queue [bounce rays | shadow rays] into eight regions (one work counter per XCD), each the x-th eighth of every sub-frame's share
    of each part, rounded up at every level (traverse_wide.h XcdRegions).  Whatever the queue lengths, the chunk size, the frames per pass and the number of waves, every ray must be
    walked exactly once: image and ray counts equal those of the static split (persistent = 0), which has no regions, no counters and no rounding."""
    w, h = size
    sc = mrt.DragonScene((w, h))
    ref = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    ref.set_option("persistent", 0); ref.set_option("frame_batch", 8)
    ref.draw(11, wait=True)
    want, counts = ref.accumulation().copy(), (ref.stats.closest_rays, ref.stats.shadow_rays)
    ref.close()
    for chunk, fb, slots, xcd in ((64, 1, 96, 1), (64, 3, 7, 1), (128, 8, 1000, 1), (192, 11, 33, 1), (320, 32, 8, 1), (64, 5, 9, 0), (4096, 8, 64, 1)):
        r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
        r.set_option("persistent", 1); r.set_option("persist_chunk", chunk); r.set_option("frame_batch", fb); r.set_option("wave_slots", slots); r.set_option("xcd_counters", xcd)
        r.draw(11, wait=True)
        assert (r.stats.closest_rays, r.stats.shadow_rays) == counts, (size, chunk, fb, slots, xcd)
        assert np.array_equal(r.accumulation().view(np.uint32), want.view(np.uint32)), (size, chunk, fb, slots, xcd)
        r.close()


# ---------------------------------------------------------------- more edge cases
def test_empty_scene_and_tiny_sizes(mrt, orc, gpu_ctx):
    class Empty(mrt.Scene):
        pass
    r = mrt.Renderer((16, 8), Empty((16, 8)), ctx=gpu_ctx)                 # no geometry: every ray misses
    r.draw(3, wait=True)
    a = r.accumulation()
    assert (a[..., :3] == 0).all() and (a[..., 3] == 1).all()
    assert (r.stats.closest_rays, r.stats.shadow_rays, r.stats.primary_rays) == (3 * 128, 0, 3 * 128)
    r.close()
    for (w, h) in [(1, 1), (3, 2), (8, 8), (9, 1)]:
        sc = mrt.CornellScene((w, h))
        r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
        r.draw(2, wait=True)
        ref, cnt = oracle_render(orc, mrt, sc, w, h, 2)
        assert np.array_equal(r.accumulation(), ref) and (r.stats.closest_rays, r.stats.shadow_rays) == cnt
        r.close()


def test_many_bounces_and_light_update(mrt, orc, gpu_ctx):
    """max_bounces up to the Halton table's limit (19: dimension 2 + 5*18 + 4 = 96 < 100) and
    mrt_scene_set_lights after the commit."""
    w, h = 64, 40
    sc = mrt.CornellScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=19)
    r.draw(2, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 2, bounces=19)
    assert_parity(r.accumulation(), ref, exact_frac=1.0)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    r.close()
    with pytest.raises(mrt.MRTError):
        mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=20)
    sc2 = mrt.CornellScene((w, h))
    r = mrt.Renderer((w, h), sc2, ctx=gpu_ctx)
    sc2.lights = [mrt.Light.pointLight([0.3, 1.5, 0.2], [2, 3, 4]), mrt.Light.sunLight([0.2, -1, 0.1], [0.5, 0.5, 0.5])]
    r.device_scene.set_lights(sc2.lights)                                  # re-upload on a committed scene
    r.draw(2, wait=True)
    ref, _ = oracle_render(orc, mrt, sc2, w, h, 2)
    assert_parity(r.accumulation(), ref, exact_frac=1.0)
    r.close()


def test_two_renderers_share_a_scene_and_resume_from_buffer(mrt, orc, gpu_ctx):
    """Checkpoint/resume of the accumulation: frames 0-1 on one renderer, its buffer copied into a
    second renderer that continues at frame index 2 — identical to 4 uninterrupted frames."""
    import torch
    w, h = 96, 64
    sc = mrt.CornellScene((w, h))
    a = mrt.Renderer((w, h), sc, ctx=gpu_ctx); a.draw(2, wait=True)
    buf = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
    a.copy_accum_to(buf.data_ptr(), buf.numel() * 4); a.wait()
    b = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    b.write_accum_from(buf.data_ptr(), buf.numel() * 4); b.frameIndex = 2
    b.draw(2, wait=True)
    ref, _ = oracle_render(orc, mrt, sc, w, h, 4)
    assert np.array_equal(b.accumulation(), ref)
    a.close(); b.close()


def test_animated_transform_recommit(mrt, orc, gpu_ctx):
    """Move an instance (new transform + re-commit = on-device rebuild) and keep rendering with the same
    renderer; the result must equal the oracle on the moved scene."""
    w, h = 128, 80
    sc = mrt.CornellScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    r.draw(1, wait=True)
    sphere = sc.models[-1]
    moved = mrt.make_transform([0.35, 0.6, -0.2], [0.3, 0.2, 0.1], 0.4)
    r.device_scene.set_instance_transform(len(sc.meshes) - 1, moved)
    with pytest.raises(mrt.MRTError):
        r.draw(1)                                  # scene modified and not re-committed
    r.device_scene.commit()
    r.frameIndex = 0
    r.draw(2, wait=True)
    sphere.meshes[0].transform = moved
    ref, _ = oracle_render(orc, mrt, sc, w, h, 2)
    assert_parity(r.accumulation(), ref, exact_frac=1.0)
    r.close()


def test_large_leaf_option_keeps_the_wide_layout(mrt, orc, gpu_ctx):
    """max_leaf > 4 cannot be addressed by the wide layout's 32-bit triangle mask (8 leaf children x 4): with the 8-wide layout the leaf limit is 4 whatever the option says
    (round 6; such a scene used to lose the layout and render on the rope kernels); a scene built without the layout (wide = 0) takes the larger leaves.  Same image."""
    w, h = 160, 90
    sc = mrt.DragonScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"max_leaf": 7})
    assert r.device_scene.stats.wide_layout == 1 and r.device_scene.stats.max_leaf_tris == 4
    r.draw(2, wait=True)
    r7 = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"max_leaf": 7, "wide": 0})
    assert r7.device_scene.stats.wide_layout == 0 and r7.device_scene.stats.max_leaf_tris == 7
    r7.draw(2, wait=True)
    assert np.array_equal(r.accumulation().view(np.uint32), r7.accumulation().view(np.uint32))
    r7.close()
    ref, _ = oracle_render(orc, mrt, sc, w, h, 2)
    assert_parity(r.accumulation(), ref)
    r.close()


# ---------------------------------------------------------------- BASELINE configs[2..4] at their stated workloads
def test_c3_64_frames_4_bounces_against_oracle(mrt, orc, gpu_ctx):
    """BASELINE configs[2]: 64 accumulated frames, 4-bounce diffuse (Halton dimensions up to 21), at a size the oracle
    finishes in seconds.  The stated bar for spp 64 is RMSE <= 2e-3; the image is expected to be bit-identical."""
    w, h = 192, 108
    sc = mrt.DragonScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=4)
    r.draw(64, wait=True)
    assert r.frameIndex == 64
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 64, bounces=4)
    assert_parity(r.accumulation(), ref, rmse_tol=2e-3)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    r.close()


def test_c3_1080p_spp64_property(mrt, gpu_ctx, dragon1080):
    """configs[2] at full size: 64 frames in one draw call == 64 frames drawn as 16 + 48 through a different batch size (the running
    average is applied in frame order whatever the batching), and the ray count is 64 frames' worth."""
    sc, r, img = dragon1080
    a = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx, max_bounces=4); a.draw(64, wait=True)
    b = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx, max_bounces=4); b.set_option("frame_batch", 2); b.draw(16, wait=True); b.draw(48, wait=True)
    ia, ib = a.accumulation(), b.accumulation()
    assert np.array_equal(ia, ib) and np.isfinite(ia).all()
    sa, sb = a.stats, b.stats
    assert (sa.closest_rays, sa.shadow_rays, sa.primary_rays) == (sb.closest_rays, sb.shadow_rays, sb.primary_rays)
    assert sa.primary_rays == 64 * 1920 * 1080 and sa.closest_rays <= 4 * sa.primary_rays
    a.close(); b.close()


@pytest.fixture(scope="module")
def garden4k(mrt, gpu_ctx):
    sc = mrt.GardenScene((3840, 2160))
    r = mrt.Renderer((3840, 2160), sc, ctx=gpu_ctx)
    r.draw(1, wait=True)
    yield sc, r, r.accumulation().copy()
    r.close()


def test_c4_4k_tiles_of_8_shards_sum_to_the_full_frame(mrt, gpu_ctx, garden4k):
    """BASELINE configs[3]: 3840x2160 tiled across 8 GPUs — here the 8 shards run one after the other on one GPU; their
    zero-initialised accumulation buffers must add up to the full frame bit for bit (what the RCCL reduce computes), and
    the shards' ray counts must add up to the full frame's (ray-count conservation)."""
    sc, r, img = garden4k
    st = r.stats
    acc = np.zeros_like(img); closest = shadow = primary = 0
    for rank in range(8):
        q = mrt.Renderer((3840, 2160), sc, ctx=gpu_ctx); q.set_shard(rank, 8); q.draw(1, wait=True)
        a = q.accumulation()
        assert (a[..., 3].sum() > 0)
        acc += a
        closest += q.stats.closest_rays; shadow += q.stats.shadow_rays; primary += q.stats.primary_rays
        q.close()
    assert np.array_equal(acc, img)
    assert (closest, shadow, primary) == (st.closest_rays, st.shadow_rays, st.primary_rays) and primary == 3840 * 2160


def test_c4_4k_determinism_and_oracle_crop(mrt, orc, gpu_ctx, garden4k):
    """Same seed, different tree (Karras, 2-triangle leaves) -> the same 4K image; and the oracle on 1/389 of the 8x8 tiles."""
    sc, r, img = garden4k
    r2 = mrt.Renderer((3840, 2160), sc, ctx=gpu_ctx, scene_options={"builder": 0, "max_leaf": 2}); r2.draw(1, wait=True)
    assert np.array_equal(r2.accumulation(), img)
    r2.close()
    ref, _ = oracle_render(orc, mrt, sc, 3840, 2160, 1, shard=(0, 389))
    ys, xs = np.mgrid[0:2160, 0:3840]
    own = (((ys // 8) * 480 + xs // 8) % 389) == 0
    assert own.sum() > 20000
    assert_parity(img[own][None], ref[own][None])
    assert (img[..., :3].sum(-1) > 0).mean() > 0.2


def test_c5_instanced_dragon_small_against_oracle(mrt, orc, gpu_ctx):
    """BASELINE configs[4] geometry (dragon x4, 3.49 M triangles) at a size the oracle finishes in seconds."""
    w, h = 320, 180
    sc = mrt.InstancedDragonScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    assert r.device_scene.stats.triangles == 885194 + 3 * 871414 and r.device_scene.stats.instances == 10
    r.draw(2, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 2)
    assert_parity(r.accumulation(), ref)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    r.close()


def test_c5_1080p_spp16_builder_invariance_and_determinism(mrt, gpu_ctx):
    """configs[4] at its stated workload (1920x1080, 16 accumulated frames): two different trees give the same image bit for bit,
    a second run reproduces it, another seed does not, and the ray counts are conserved."""
    sc = mrt.InstancedDragonScene((1920, 1080))
    a = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx); a.draw(16, wait=True); ia = a.accumulation()
    b = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx, scene_options={"builder": 0, "max_leaf": 2}); b.draw(16, wait=True)
    assert np.array_equal(b.accumulation(), ia)
    assert (a.stats.closest_rays, a.stats.shadow_rays) == (b.stats.closest_rays, b.stats.shadow_rays)
    b.close()
    c = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx); c.set_option("frames_in_flight", 1); c.set_option("frame_batch", 1); c.draw(16, wait=True)
    assert np.array_equal(c.accumulation(), ia)
    c.close()
    d = mrt.Renderer((1920, 1080), sc, ctx=gpu_ctx, seed=5); d.draw(16, wait=True)
    assert not np.array_equal(d.accumulation(), ia)
    d.close()
    st = a.stats
    assert st.primary_rays == 16 * 1920 * 1080 and st.primary_rays <= st.closest_rays <= 3 * st.primary_rays and 0 < st.shadow_rays <= st.closest_rays
    assert np.isfinite(ia).all() and (ia[..., 3] == 1).all()
    a.close()


# ---------------------------------------------------------------- the one-launch-per-frame megakernel (renderer option megakernel = 1)
@pytest.mark.parametrize("case", ["cornell", "dragon", "lights", "ragged_shard"])
def test_megakernel_matches_oracle_and_wavefront_pipeline(mrt, orc, gpu_ctx, case):
    """k_megakernel carries whole paths per lane (no ray queues, no k_shade, no k_accumulate): same image, same ray counts, bit for bit."""
    bounces, shard = 3, None
    if case == "cornell":
        w, h, frames = 256, 256, 3; sc = mrt.CornellScene((w, h))
    elif case == "dragon":
        w, h, frames = 320, 180, 2; sc = mrt.DragonScene((w, h)); bounces = 4
    elif case == "lights":
        w, h, frames = 160, 96, 2; sc = mrt.GardenScene((w, h))
        sc.lights = sc.lights + [mrt.Light.pointLight([0, 2.5, 1], [3, 2, 1]), mrt.Scene.setupLight()]
    else:
        w, h, frames = 75, 43, 3; sc = mrt.CornellScene((w, h)); shard = (1, 3)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=bounces)
    r.set_option("megakernel", 1)
    if shard: r.set_shard(*shard)
    r.draw(frames, wait=True)
    ref, cnt = oracle_render(orc, mrt, sc, w, h, frames, bounces=bounces, shard=shard)
    assert_parity(r.accumulation(), ref, exact_frac=1.0)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == cnt
    q = mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=bounces)
    if shard: q.set_shard(*shard)
    q.draw(frames, wait=True)
    assert np.array_equal(q.accumulation(), r.accumulation()) and q.stats.primary_rays == r.stats.primary_rays
    # switching the option between draws continues the same accumulation
    r.set_option("megakernel", 0); r.draw(2, wait=True); q.set_option("megakernel", 1); q.draw(2, wait=True)
    assert np.array_equal(q.accumulation(), r.accumulation()) and r.frameIndex == frames + 2
    r.close(); q.close()
