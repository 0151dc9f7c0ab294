"""Two-level scenes: one BLAS per distinct mesh shared by its instances + a TLAS (scene option instancing = 1),
against (a) the oracle's two-level restatement — bit for bit — and (b) the flattened scene — within the stated tolerance, since the
triangle test then runs in object space with different roundings."""
import time

import numpy as np
import pytest

from test_gpu_parity import assert_parity, TOL_ABS


def _scene(mrt, size):
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [
                mrt.Model(name="plane", position=[0, 0, 0], scale=10),
                mrt.Model(name="sphere", position=[-1.2, 0.0, 0.4], scale=1),
                mrt.Model(name="sphere", position=[1.4, 0.0, -0.2], rotation=[0.3, 1.1, -0.4], scale=1.7),
                mrt.Model(name="teapot", position=[0.1, 0.0, 1.6], rotation=[0, 0.6, 0], scale=0.01),
                mrt.Model(name="teapot", position=[-0.6, 0.0, -1.0], rotation=[0, -1.3, 0], scale=0.013),
                mrt.Model(name="sphere", position=[0.2, 1.1, 0.8], scale=0.4),
            ]
    return S(size)


def _rays(rng, n):
    o = np.array([0.0, 1.2, 4.5]) + rng.normal(size=(n, 3)) * 0.4
    t = np.c_[rng.uniform(-2.5, 2.5, n), rng.uniform(0, 1.8, n), rng.uniform(-1.5, 2.0, n)]
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32); r[:, 0:3] = o; r[:, 4:7] = d; r[:, 7] = np.inf
    return r


def test_shared_geometry_is_detected_and_oracle_instancing_matches_flattening(mrt, orc):
    sc = _scene(mrt, (96, 64))
    shared = mrt.flatten_scene(sc, share=True)
    assert [e[4] for e in shared] == [-1, -1, 1, -1, 3, 1]            # spheres are instances of mesh 1, the second teapot of mesh 3
    flat = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    two = orc.OracleScene(shared, sc.lights, instancing=True)
    assert flat.triangles == two.triangles
    rays = _rays(np.random.default_rng(5), 4000)
    a, b = flat.intersect_closest(rays), two.intersect_closest(rays)
    bb = two.intersect_closest(rays, brute=True)
    for f in a.dtype.names:
        assert np.array_equal(b[f], bb[f]), f                        # the per-BLAS BVH is exact
    same = (a["type"] == b["type"]) & (a["instance_id"] == b["instance_id"]) & (a["geometry_id"] == b["geometry_id"]) & (a["primitive_id"] == b["primitive_id"])
    assert same.mean() > 0.998                                        # object-space vs world-space rounding may flip a grazing hit
    hit = same & (a["type"] == 1)
    assert hit.sum() > 1000
    assert np.allclose(a["distance"][hit], b["distance"][hit], rtol=2e-5, atol=1e-6)
    assert np.abs(a["u"][hit] - b["u"][hit]).max() < 1e-3 and np.abs(a["v"][hit] - b["v"][hit]).max() < 1e-3
    rays[:, 7] = 3.0
    assert (flat.intersect_any(rays) == two.intersect_any(rays)).mean() > 0.998
    # images: the formal bar of SURVEY §8(d)
    w, h = 96, 64
    fa = orc.OracleRenderer(flat, w, h, camera=sc.camera); fa.render(2)
    fb = orc.OracleRenderer(two, w, h, camera=sc.camera); fb.render(2)
    d = np.abs(fa.accumulation()[..., :3].astype(np.float64) - fb.accumulation()[..., :3])
    assert (d.max(-1) <= TOL_ABS).mean() >= 0.995 and np.sqrt((d ** 2).sum(-1).mean()) <= 2e-3
    ca, cb = fa.counters(), fb.counters()
    assert abs(ca[0] - cb[0]) <= 0.002 * ca[0] and abs(ca[1] - cb[1]) <= 0.002 * ca[1]


@pytest.mark.gpu
def test_two_level_gpu_matches_two_level_oracle_bit_for_bit(mrt, orc, gpu_ctx):
    w, h = 160, 96
    sc = _scene(mrt, (w, h))
    two = orc.OracleScene(mrt.flatten_scene(sc, share=True), sc.lights, instancing=True)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"instancing": 1})
    st = r.device_scene.stats
    assert st.instances == 6 and st.triangles == two.triangles
    rays = _rays(np.random.default_rng(6), 20000)
    g, o = r.device_scene.intersect_closest(rays), two.intersect_closest(rays)
    for f in ("type", "distance", "instance_id", "geometry_id", "primitive_id", "u", "v"):
        assert np.array_equal(g[f], o[f]), f
    # the render kernels' own walk (8-wide TLAS + 8-wide BLASes, one stream loop) answers the same queries with the same bits
    assert r.device_scene.stats.bvh_nodes > 0
    gs = r.device_scene.intersect_stream(rays)
    for f in ("type", "distance", "instance_id", "geometry_id", "primitive_id", "u", "v"):
        assert np.array_equal(gs[f], o[f]), f
    rays[:, 7] = 3.0
    assert np.array_equal(r.device_scene.intersect_any(rays), two.intersect_any(rays))
    assert np.array_equal(r.device_scene.intersect_stream(rays, any_hit=True)["type"], two.intersect_any(rays))
    r.draw(3, wait=True)
    ref = orc.OracleRenderer(two, w, h, camera=sc.camera); ref.render(3)
    assert_parity(r.accumulation(), ref.accumulation())
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    # the rope kernels (wide_bounce = 0: traverse_instanced.h) render the same image
    rope = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"instancing": 1}); rope.set_option("wide_bounce", 0)
    rope.draw(3, wait=True)
    assert np.array_equal(rope.accumulation(), r.accumulation())
    rope.close()
    # against the flattened scene on the GPU: the formal tolerance
    f = mrt.Renderer((w, h), sc, ctx=gpu_ctx); f.draw(3, wait=True)
    d = np.abs(f.accumulation()[..., :3].astype(np.float64) - r.accumulation()[..., :3])
    assert (d.max(-1) <= TOL_ABS).mean() >= 0.995
    f.close(); r.close()


def _swarm(mrt, size, n, seed):
    rng = np.random.default_rng(seed)
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [mrt.Model(name="plane", position=[0, 0, 0], scale=10)]
            for i in range(n):
                name = "sphere" if i % 3 else "teapot"
                sc = float(rng.uniform(0.08, 0.3)) * (1.0 if name == "sphere" else 0.012)
                self.models.append(mrt.Model(name=name, position=[float(rng.uniform(-2.5, 2.5)), float(rng.uniform(0.1, 2.0)), float(rng.uniform(-2.0, 2.0))],
                                             rotation=[float(x) for x in rng.uniform(-3, 3, 3)], scale=sc))
    return S(size)


@pytest.mark.gpu
def test_many_instances_deep_tlas(mrt, orc, gpu_ctx):
    """150 instances of two meshes: a TLAS of several 8-wide levels, overlapping instance boxes, rays that enter and leave many BLASes —
    the stream traversal (parked TLAS groups, world rays in LDS) against the two-level oracle bit for bit; then every transform changes
    and only the TLAS is rebuilt."""
    w, h = 128, 80
    sc = _swarm(mrt, (w, h), 150, 21)
    shared = mrt.flatten_scene(sc, share=True)
    two = orc.OracleScene(shared, sc.lights, instancing=True)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"instancing": 1})
    assert r.device_scene.stats.instances == 151
    rays = _rays(np.random.default_rng(8), 30000)
    o = two.intersect_closest(rays)
    assert (o["type"] == 1).mean() > 0.5 and len(np.unique(o["instance_id"])) > 60
    for g in (r.device_scene.intersect_stream(rays), r.device_scene.intersect_closest(rays)):
        for f in ("type", "distance", "instance_id", "geometry_id", "primitive_id", "u", "v"):
            assert np.array_equal(g[f], o[f]), f
    rays[:, 7] = 2.5
    assert np.array_equal(r.device_scene.intersect_stream(rays, any_hit=True)["type"], two.intersect_any(rays))
    r.draw(2, wait=True)
    ref = orc.OracleRenderer(two, w, h, camera=sc.camera); ref.render(2)
    assert_parity(r.accumulation(), ref.accumulation())
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    # move everything
    rng = np.random.default_rng(22)
    blas_ms = r.device_scene.stats.build_ms
    for i in range(1, 151):
        xf = mrt.make_transform([float(rng.uniform(-2.5, 2.5)), float(rng.uniform(0.1, 2.0)), float(rng.uniform(-2.0, 2.0))], [float(x) for x in rng.uniform(-3, 3, 3)],
                                float(rng.uniform(0.08, 0.3)) * (1.0 if i % 3 != 1 else 0.012))
        r.device_scene.set_instance_transform(i, xf); two.set_transform(i, xf.reshape(16))
    r.device_scene.commit()
    assert r.device_scene.stats.build_ms == blas_ms
    rays[:, 7] = np.inf
    o = two.intersect_closest(rays); g = r.device_scene.intersect_stream(rays)
    for f in ("type", "distance", "instance_id", "geometry_id", "primitive_id", "u", "v"):
        assert np.array_equal(g[f], o[f]), f
    r.close()


@pytest.mark.gpu
def test_many_tiny_instances(mrt, orc, gpu_ctx):
    """600 instances of the two-triangle plane (entered without a root-node test: their packets are the pending set at once) among 40 spheres: a TLAS
    of four 8-wide levels whose leaves are mostly such tiny BLASes; queries and an image against the two-level oracle, bit for bit."""
    rng = np.random.default_rng(31)
    w, h = 96, 64
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            self.models = [mrt.Model(name="plane", position=[0, 0, 0], scale=10)]
            for i in range(640):
                name = "sphere" if i % 16 == 0 else "plane"
                self.models.append(mrt.Model(name=name, position=[float(rng.uniform(-2.5, 2.5)), float(rng.uniform(0.05, 2.2)), float(rng.uniform(-2.0, 2.5))],
                                             rotation=[float(x) for x in rng.uniform(-3, 3, 3)], scale=float(rng.uniform(0.05, 0.25))))
    sc = S((w, h))
    two = orc.OracleScene(mrt.flatten_scene(sc, share=True), sc.lights, instancing=True)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"instancing": 1})
    assert r.device_scene.stats.instances == 641
    rays = _rays(np.random.default_rng(9), 20000)
    o = two.intersect_closest(rays); g = r.device_scene.intersect_stream(rays)
    assert (o["type"] == 1).mean() > 0.5 and len(np.unique(o["instance_id"])) > 200
    for f in ("type", "distance", "instance_id", "geometry_id", "primitive_id", "u", "v"):
        assert np.array_equal(g[f], o[f]), f
    rays[:, 7] = 2.0
    assert np.array_equal(r.device_scene.intersect_stream(rays, any_hit=True)["type"], two.intersect_any(rays))
    r.draw(2, wait=True)
    ref = orc.OracleRenderer(two, w, h, camera=sc.camera); ref.render(2)
    assert_parity(r.accumulation(), ref.accumulation())
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    r.close()


@pytest.mark.gpu
def test_two_level_with_materials_and_four_bounces(mrt, orc, gpu_ctx):
    """The two extensions together: instanced scene (8-wide TLAS + BLASes) with the materials path of k_shade (sphere.mtl carries Ks / Ns, so the
    specular lobe and the lobe-sorted queues are exercised), 4 bounces, against the two-level oracle with its materials switch."""
    w, h = 128, 80
    sc = _scene(mrt, (w, h))
    two = orc.OracleScene(mrt.flatten_scene(sc, share=True), sc.lights, instancing=True)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=4, scene_options={"instancing": 1})
    r.set_option("materials", 1)
    r.draw(3, wait=True)
    ref = orc.OracleRenderer(two, w, h, max_bounces=4, camera=sc.camera); ref.set_materials(True); ref.render(3)
    assert_parity(r.accumulation(), ref.accumulation())
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    r.close()


@pytest.mark.gpu
def test_transform_change_rebuilds_only_the_tlas(mrt, orc, gpu_ctx):
    """Animated transforms (SURVEY §8 f-3 'refit'): set_instance_transform + commit on a two-level scene leaves the BLASes alone."""
    w, h = 128, 80
    sc = _scene(mrt, (w, h))
    two = orc.OracleScene(mrt.flatten_scene(sc, share=True), sc.lights, instancing=True)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"instancing": 1})
    blas_ms = r.device_scene.stats.build_ms
    xf = mrt.make_transform([1.0, 0.3, 0.5], [0.2, 2.0, 0.1], 1.2)
    t0 = time.perf_counter()
    r.device_scene.set_instance_transform(2, xf); r.device_scene.commit()
    dt = (time.perf_counter() - t0) * 1e3
    assert r.device_scene.stats.build_ms == blas_ms                   # no BLAS was rebuilt
    assert dt < 20.0, f"TLAS update took {dt:.2f} ms"
    two.set_transform(2, xf.reshape(16))
    r.frameIndex = 0; r.draw(2, wait=True)
    ref = orc.OracleRenderer(two, w, h, camera=sc.camera); ref.render(2)
    got = r.accumulation()
    # the renderer kept accumulating into its previous target: frame 0 overwrites it (Raytracing.metal:395), so this is a fresh image
    assert_parity(got, ref.accumulation())
    r.close()


@pytest.mark.gpu
def test_c5_as_one_blas_times_four_instances(mrt, orc, gpu_ctx):
    """BASELINE configs[4] built the instanced way: ONE dragon BLAS shared by four instances (+ the rest of DragonScene)."""
    w, h = 320, 180
    sc = mrt.InstancedDragonScene((w, h))
    shared = mrt.flatten_scene(sc, share=True)
    assert sum(1 for e in shared if e[4] >= 0) == 4                    # three extra dragons + the second sphere
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"instancing": 1})
    st = r.device_scene.stats
    assert st.triangles == 885194 + 3 * 871414 and st.instances == 10
    flat_bytes = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"rope": 1})      # like for like: the BLASes of a two-level scene keep both layouts (8-wide + rope)
    assert st.scene_bytes < 0.45 * flat_bytes.device_scene.stats.scene_bytes      # one dragon's worth of BLAS, not four
    flat_bytes.close()
    r.draw(2, wait=True)
    two = orc.OracleScene(shared, sc.lights, instancing=True)
    ref = orc.OracleRenderer(two, w, h, camera=sc.camera); ref.render(2)
    assert_parity(r.accumulation(), ref.accumulation())
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    r.close()


@pytest.mark.gpu
def test_flattened_instances_equal_separately_added_meshes(mrt, orc, gpu_ctx):
    """instancing = 0 with mrt_scene_add_instance (shared host arrays) is the same flattened scene bit for bit."""
    w, h = 96, 64
    sc = _scene(mrt, (w, h))
    a = mrt.Renderer((w, h), sc, ctx=gpu_ctx); a.draw(2, wait=True)
    flat = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    ref = orc.OracleRenderer(flat, w, h, camera=sc.camera); ref.render(2)
    assert_parity(a.accumulation(), ref.accumulation(), exact_frac=1.0)
    a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("instancing", [0, 1])
def test_commit_validates_every_index_of_the_layout(mrt, gpu_ctx, instancing):
    """mrt_scene_commit checks that every child / packet / instance index of the 8-wide layout lies inside its array (the class of error behind
    a GPU memory fault seen once on a work-in-progress tree): a broken word is refused with a message, not dereferenced."""
    import ctypes as C
    from metal_raytracing_amd._ffi import lib
    sc = _scene(mrt, (64, 48))
    ds = mrt.DeviceScene(gpu_ctx, sc, {"instancing": instancing})
    assert lib.mrt_debug_validate(ds.handle) == 0
    # word 4 = child_base, word 5 = tri_base of node 0: the flattened scene's root has internal children, the 6-instance TLAS root only leaf children
    cases = [(0, 5, 0x7FFFFFF0, "outside")] if instancing else [(0, 4, 0x00FFFFF0, "outside"), (0, 4, 0, "after their parent")]
    for node, word, value, what in cases:
        rc = lib.mrt_debug_validate_patched(ds.handle, node, word, value)          # the validator on a copy of the nodes with that word replaced: the scene's own arrays stay as committed
        msg = lib.mrt_last_error().decode()
        assert rc == 5 and "validation failed" in msg and what in msg, (rc, msg)
    assert lib.mrt_debug_validate(ds.handle) == 0
    ds.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["six_instances", "dragon4"])
def test_binned_two_level_walk_is_the_one_loop_walk(mrt, gpu_ctx, scene_name):
    """Bounce and shadow rays of a two-level scene as TLAS pass + BLAS pass over (ray, instance) pairs (tl_pairs = 1, default) against the one-loop walk of both levels
    (tl_pairs = 0): same image bit for bit, same ray counts; also with a pair queue so small that almost every push is refused and walks its instance in place, with
    one-frame passes, and with four-frame passes on three lanes (the accumulate's zeroing of the pair counters)."""
    w, h = (160, 96) if scene_name == "six_instances" else (256, 144)
    sc = _scene(mrt, (w, h)) if scene_name == "six_instances" else mrt.InstancedDragonScene((w, h))
    imgs = {}
    # binned (default): the tree-less TLAS pass; tl_pairs = 2: the stream walk of the 8-wide TLAS; fuse_primary = 0: bounce 0 shades the hit records of a primary launch
    for name, opts in (("one_loop", {"tl_pairs": 0}), ("binned", {}), ("binned_tiny_queue", {"tl_pair_cap": 257}), ("binned_unpacked_shade", {"shade_pack": 0}), ("binned_primary_launch", {"fuse_primary": 0}),
                       ("binned_tlas_walk", {"tl_pairs": 2}), ("binned_tlas_walk_tiny_queue", {"tl_pairs": 2, "tl_pair_cap": 600}), ("binned_one_frame_passes", {"frame_batch": 1}),
                       ("binned_three_lanes", {"frame_batch": 4, "frames_in_flight": 3})):
        r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"instancing": 1})
        for k, v in opts.items(): r.set_option(k, v)
        r.draw(5, wait=True); r.draw(20, wait=True)
        imgs[name] = (r.accumulation().copy(), r.stats.closest_rays, r.stats.shadow_rays)
        r.close()
    for name, v in imgs.items():
        assert np.array_equal(v[0].view(np.uint32), imgs["one_loop"][0].view(np.uint32)), name
        assert v[1:] == imgs["one_loop"][1:], name


@pytest.mark.gpu
def test_megakernel_refuses_what_it_cannot_render(mrt, gpu_ctx):
    """megakernel = 1 is the one-launch-per-frame mode of flattened scenes on the 8-wide layout: a two-level scene, the materials extension or a scene without that layout
    get MRT_ERR_UNSUPPORTED and a message naming the reason — not a silent fall-back to the pipeline — and the renderer keeps working once the option is cleared."""
    from metal_raytracing_amd._ffi import MRTError
    w, h = 64, 48
    sc = _scene(mrt, (w, h))
    for sopt, ropt, why in (({"instancing": 1}, {}, "two-level"), ({}, {"materials": 1}, "materials"), ({"wide": 0}, {}, "8-wide")):
        r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options=sopt)
        for k, v in ropt.items(): r.set_option(k, v)
        r.set_option("megakernel", 1)
        with pytest.raises(MRTError) as e:
            r.draw(1, wait=True)
        assert why in str(e.value) and "megakernel" in str(e.value), str(e.value)
        r.set_option("megakernel", 0); r.draw(2, wait=True)
        assert np.isfinite(r.accumulation()).all() and r.stats.frames == 2
        r.close()


@pytest.mark.gpu
def test_flattened_scene_recommits_new_transforms_without_its_geometry(mrt, gpu_ctx):
    """An animated FLATTENED scene: mrt_scene_set_instance_transform + mrt_scene_commit rebuilds the world-space BVH from the geometry the previous commit left on the
    device (no staging, no upload of positions / normals / indices).  The image must be the one a scene created with those transforms gives, bit for bit, twice in a
    row, and any other change of the scene (an option) must take the full path again."""
    import ctypes as C
    w, h = 160, 96
    sc = _scene(mrt, (w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    r.draw(2, wait=True)
    first = (C.c_double * 6)(); mrt._ffi.check(mrt.lib.mrt_debug_commit_times(r.device_scene.handle, first))
    for step, (mesh_id, pos) in enumerate([(2, [1.0, 0.3, -0.8]), (4, [-0.9, 0.2, 0.7])]):
        xf = np.eye(4, dtype=np.float32); xf[:3, 3] = pos; xf[0, 0] = xf[1, 1] = xf[2, 2] = 0.8 if mesh_id == 2 else 0.012
        xf = np.ascontiguousarray(xf.T)                                # (4,4) [col][row], as Mesh.transform
        r.device_scene.set_instance_transform(mesh_id, xf); r.device_scene.commit()
        again = (C.c_double * 6)(); mrt._ffi.check(mrt.lib.mrt_debug_commit_times(r.device_scene.handle, again))
        sc.meshes[mesh_id].transform = xf.reshape(4, 4)
        r.drawableSizeWillChange((w, h)); r.draw(3, wait=True)
        fresh = mrt.Renderer((w, h), sc, ctx=gpu_ctx); fresh.draw(3, wait=True)
        assert np.array_equal(r.accumulation().view(np.uint32), fresh.accumulation().view(np.uint32)), step
        assert (r.stats.closest_rays, r.stats.shadow_rays) == (fresh.stats.closest_rays, fresh.stats.shadow_rays)
        a, b = r.device_scene.stats, fresh.device_scene.stats
        assert (a.triangles, a.bvh_nodes, a.bvh_leaves, a.wide_depth, a.scene_bytes) == (b.triangles, b.bvh_nodes, b.bvh_leaves, b.wide_depth, b.scene_bytes)
        fresh.close()
        print("commit phases, first:", [round(x, 3) for x in first], "transform-only:", [round(x, 3) for x in again])
    mrt._ffi.check(mrt.lib.mrt_scene_set_option(r.device_scene.handle, b"presplit", 0.0)); r.device_scene.commit()      # not a transform change: full path
    r.drawableSizeWillChange((w, h)); r.draw(3, wait=True)
    fresh = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"presplit": 0}); fresh.draw(3, wait=True)
    assert np.array_equal(r.accumulation().view(np.uint32), fresh.accumulation().view(np.uint32))
    fresh.close(); r.close()


@pytest.mark.gpu
def test_primary_hint_on_two_level_scenes(mrt, orc, gpu_ctx):
    """The hint of two-level scenes — (packet | instance << 24) of the pixel's last primary hit, tested first in that instance's object space — changes no pixel:
    with it and without it, over frames that reuse it, after the instances have moved (stale hints: legal guesses or rejected), and against the oracle."""
    w, h = 160, 96
    sc = _scene(mrt, (w, h))
    opts = {"instancing": 1}
    imgs = {}
    for hint in (1, 0):
        r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options=opts)
        r.set_option("primary_hint", hint)
        r.draw(5, wait=True)
        xf = np.eye(4, dtype=np.float32); xf[:3, 3] = [1.0, 0.3, -0.8]; xf = np.ascontiguousarray(xf.T)
        r.device_scene.set_instance_transform(2, xf); r.device_scene.commit()
        r.draw(4, wait=True)
        imgs[hint] = (r.accumulation().copy(), r.stats.closest_rays, r.stats.shadow_rays)
        r.close()
    assert np.array_equal(imgs[1][0].view(np.uint32), imgs[0][0].view(np.uint32)) and imgs[1][1:] == imgs[0][1:]
    # and both are the oracle's image of the same sequence
    two = orc.OracleScene(mrt.flatten_scene(sc, share=True), sc.lights, instancing=True)
    ref = orc.OracleRenderer(two, w, h, camera=sc.camera); ref.render(5)
    two.set_transform(2, xf.reshape(16)); ref.render(4)
    assert_parity(imgs[1][0], ref.accumulation(), exact_frac=1.0)
    assert imgs[1][1:] == ref.counters()
