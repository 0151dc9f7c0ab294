"""Deforming geometry (SURVEY §8 f-3 "+ refit"): mrt_scene_update_mesh + commit refits the 8-wide tree of a flattened scene — packets rewritten, boxes recomputed bottom-up,
the tree's shape kept.  The closest hit does not depend on the tree, so the refitted scene must render what a fresh build of the deformed scene renders, and what the oracle
renders, bit for bit; its intersector must answer like brute force."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _deformed(mrt, size, amp, phase):
    """DragonScene with the dragon's vertices pushed along their normals by a wave (and the normals tilted with it)."""
    sc = mrt.DragonScene(size)
    meshes = mrt.flatten_scene(sc, share=True)
    out = []
    for k, (pos, nrm, xf, subs, source) in enumerate(meshes):
        pos = np.asarray(pos, np.float32).copy(); nrm = np.asarray(nrm, np.float32).copy()
        if len(pos) > 100000:        # the dragon
            wv = (amp * np.sin(9.0 * pos[:, 1] + phase) * np.cos(7.0 * pos[:, 0] - phase)).astype(np.float32)
            pos = (pos + nrm * wv[:, None]).astype(np.float32)
            t = (nrm + np.float32(0.3) * wv[:, None] * np.array([1, 0, 0], np.float32)).astype(np.float32)
            nrm = (t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-6)).astype(np.float32)
        out.append((pos, nrm, xf, subs, source))
    return sc, out


def test_refit_gives_the_image_of_a_fresh_build_and_of_the_oracle(mrt, orc, gpu_ctx):
    w, h = 256, 144
    sc0, base = _deformed(mrt, (w, h), 0.0, 0.0)
    big = [k for k, m in enumerate(base) if len(m[0]) > 100000]
    assert len(big) == 1
    r = mrt.Renderer((w, h), sc0, ctx=gpu_ctx)
    ds = r.device_scene
    r.draw(3, wait=True)
    build_ms = ds.stats.build_ms
    assert ds.refits == 0
    for step, (amp, phase) in enumerate([(0.02, 0.3), (0.05, 1.1)]):
        _, meshes = _deformed(mrt, (w, h), amp, phase)
        ds.update_mesh(big[0], meshes[big[0]][0], meshes[big[0]][1]); ds.commit()
        assert ds.refits == step + 1, "the commit after update_mesh alone must refit, not build"
        refit_ms = ds.stats.build_ms
        r.frameIndex = 0; r.draw(5, wait=True)
        img = r.accumulation().copy(); cnt = (r.stats.closest_rays, r.stats.shadow_rays)
        # a fresh build of the same deformed scene (scene option refit = 0 makes the same calls build)
        r2 = mrt.Renderer((w, h), sc0, ctx=gpu_ctx, scene_options={"refit": 0})
        r2.device_scene.update_mesh(big[0], meshes[big[0]][0], meshes[big[0]][1]); r2.device_scene.commit()
        assert r2.device_scene.refits == 0
        r2.draw(5, wait=True)
        assert np.array_equal(img.view(np.uint32), r2.accumulation().view(np.uint32)), "refit and fresh build must render the same image"
        osc = orc.OracleScene([m[:4] for m in meshes], sc0.lights)
        o = orc.OracleRenderer(osc, w, h, seed=1, max_bounces=3, camera=sc0.camera); o.render(5)
        assert np.array_equal(img.view(np.uint32), o.accumulation().view(np.uint32)), "and the oracle's"
        assert (r2.stats.closest_rays, r2.stats.shadow_rays) == o.counters()
        # the intersector on the refitted tree against the oracle's brute force
        rng = np.random.default_rng(7 + step)
        rays = np.zeros((3000, 8), np.float32); rays[:, 0:3] = rng.uniform([-1.5, 0.1, 0.5], [1.5, 1.5, 4.0], (3000, 3)); d = rng.normal(size=(3000, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 7] = np.inf
        g, b = ds.intersect_closest(rays), osc.intersect_closest(rays)
        for f in ("type", "primitive_id", "geometry_id", "instance_id"): assert np.array_equal(g[f], b[f]), f
        assert np.array_equal(g["distance"].view(np.uint32), b["distance"].view(np.uint32))
        r2.close()
        assert refit_ms < build_ms, (refit_ms, build_ms)
    # anything else changing with the vertices makes the next commit a build
    ds.update_mesh(big[0], base[big[0]][0], base[big[0]][1]); ds.set_instance_transform(big[0], np.asarray(base[big[0]][2], np.float32)); ds.commit()
    assert ds.refits == 0
    r.frameIndex = 0; r.draw(3, wait=True)
    ref = mrt.Renderer((w, h), sc0, ctx=gpu_ctx); ref.draw(3, wait=True)
    assert np.array_equal(r.accumulation().view(np.uint32), ref.accumulation().view(np.uint32))
    ref.close(); r.close()


@pytest.mark.parametrize("sopt,ropt", [({"rope": 1}, {"wide_bounce": 0, "primary_wide": 0}), ({"wide": 0}, {})])
def test_refit_of_the_rope_layout(mrt, orc, gpu_ctx, sopt, ropt):
    """The rope layout is refitted too (round 6): beside the 8-wide one (scene option rope = 1; the renderer's A/B switches then walk the rope copy) and alone (wide = 0: every ray
    on the rope kernels).  A leaf none of whose triangles moved keeps its box (the clipped boxes of the pre-split walls survive).  Same image as a fresh build and as the oracle."""
    w, h = 160, 90
    sc0, base = _deformed(mrt, (w, h), 0.0, 0.0)
    big = [k for k, m in enumerate(base) if len(m[0]) > 100000]
    r = mrt.Renderer((w, h), sc0, ctx=gpu_ctx, scene_options=sopt)
    for k, v in ropt.items(): r.set_option(k, v)
    ds = r.device_scene
    assert ds.stats.wide_layout == (0 if sopt.get("wide") == 0 else 1)
    r.draw(2, wait=True)
    _, meshes = _deformed(mrt, (w, h), 0.03, 0.9)
    ds.update_mesh(big[0], meshes[big[0]][0], meshes[big[0]][1]); ds.commit()
    assert ds.refits == 1 and ds.stats.refits == 1
    r.frameIndex = 0; r.reset_stats(); r.draw(4, wait=True)
    osc = orc.OracleScene([m[:4] for m in meshes], sc0.lights)
    o = orc.OracleRenderer(osc, w, h, seed=1, max_bounces=3, camera=sc0.camera); o.render(4)
    assert np.array_equal(r.accumulation().view(np.uint32), o.accumulation().view(np.uint32))
    assert (r.stats.closest_rays, r.stats.shadow_rays) == o.counters()
    rng = np.random.default_rng(3)
    rays = np.zeros((3000, 8), np.float32); rays[:, 0:3] = rng.uniform([-1.5, 0.1, 0.5], [1.5, 1.5, 4.0], (3000, 3)); d = rng.normal(size=(3000, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 7] = np.inf
    g, b = ds.intersect_closest(rays), osc.intersect_closest(rays)
    for f in ("type", "primitive_id", "geometry_id", "instance_id"): assert np.array_equal(g[f], b[f]), f
    assert np.array_equal(g["distance"].view(np.uint32), b["distance"].view(np.uint32))
    r.close()


def _deformed_instanced(mrt, size, amp, phase):
    """Dragon x 4 as ONE mesh + four instances (the shared-BLAS scene), the mesh's vertices pushed along their normals by a wave."""
    sc = mrt.InstancedDragonScene(size)
    meshes = mrt.flatten_scene(sc, share=True)
    out = []
    for (pos, nrm, xf, subs, source) in meshes:
        pos = np.asarray(pos, np.float32).copy(); nrm = np.asarray(nrm, np.float32).copy()
        if len(pos) > 100000:        # the dragon mesh and its instances (the same arrays: the oracle takes each entry's own)
            wv = (amp * np.sin(9.0 * pos[:, 1] + phase) * np.cos(7.0 * pos[:, 0] - phase)).astype(np.float32)
            pos = (pos + nrm * wv[:, None]).astype(np.float32)
            t = (nrm + np.float32(0.3) * wv[:, None] * np.array([1, 0, 0], np.float32)).astype(np.float32)
            nrm = (t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-6)).astype(np.float32)
        out.append((pos, nrm, xf, subs, source))
    return sc, out


def test_two_level_refit_gives_the_image_of_a_fresh_build_and_of_the_oracle(mrt, orc, gpu_ctx):
    """mrt_scene_update_mesh + commit on a TWO-LEVEL scene (Renderer.swift:193-213 is the instance-based API): the changed mesh's BLAS is refitted in place — the 8-wide layout the
    render kernels walk AND the rope copy the query API walks — its four instances share the result, the TLAS follows the new BLAS box.  Same image as a fresh two-level build of
    the deformed scene and as the two-level oracle, bit for bit; both intersectors answer like the oracle's brute force; MRTSceneStats shows what the refit did to the tree."""
    w, h = 192, 108
    sc0, base = _deformed_instanced(mrt, (w, h), 0.0, 0.0)
    big = [k for k, m in enumerate(base) if len(m[0]) > 100000 and m[4] < 0]          # the mesh itself; its instances (source >= 0) carry the same arrays
    assert len(big) == 1 and sum(1 for m in base if m[4] == big[0]) >= 3
    r = mrt.Renderer((w, h), sc0, ctx=gpu_ctx, scene_options={"instancing": 1})
    ds = r.device_scene
    r.draw(3, wait=True)
    st0 = ds.stats
    assert ds.refits == 0 and st0.refits == 0 and st0.wide_cost > 0 and st0.wide_cost == st0.wide_cost_built
    costs = []
    for step, (amp, phase) in enumerate([(0.02, 0.3), (0.05, 1.1)]):
        _, meshes = _deformed_instanced(mrt, (w, h), amp, phase)
        ds.update_mesh(big[0], meshes[big[0]][0], meshes[big[0]][1]); ds.commit()
        st = ds.stats
        assert ds.refits == step + 1 and st.refits == step + 1, "the commit after update_mesh alone must refit the BLAS, not build"
        assert st.build_ms < st0.build_ms, (st.build_ms, st0.build_ms)
        assert st.wide_cost_built == st0.wide_cost_built and st.wide_cost > st.wide_cost_built, "a deformed mesh on the build's tree: the boxes loosen, and the statistics say so"
        assert st.sah_cost > st0.sah_cost          # the build's figure scaled by each BLAS's own cost ratio (mean over the BLASes)
        assert st.leaf_growth > 1.2, st.leaf_growth          # the moved mesh's leaf boxes against the build's: the sharper signal
        costs.append(st.wide_cost)
        r.frameIndex = 0; r.reset_stats(); r.draw(4, wait=True)
        img = r.accumulation().copy(); cnt = (r.stats.closest_rays, r.stats.shadow_rays)
        r2 = mrt.Renderer((w, h), sc0, ctx=gpu_ctx, scene_options={"instancing": 1, "refit": 0})          # the same calls build
        r2.device_scene.update_mesh(big[0], meshes[big[0]][0], meshes[big[0]][1]); r2.device_scene.commit()
        assert r2.device_scene.refits == 0
        r2.draw(4, wait=True)
        assert np.array_equal(img.view(np.uint32), r2.accumulation().view(np.uint32)), "refit and fresh build must render the same image"
        assert cnt == (r2.stats.closest_rays, r2.stats.shadow_rays)
        osc = orc.OracleScene(meshes, sc0.lights, instancing=True)
        o = orc.OracleRenderer(osc, w, h, seed=1, max_bounces=3, camera=sc0.camera); o.render(4)
        assert np.array_equal(img.view(np.uint32), o.accumulation().view(np.uint32)), "and the two-level oracle's"
        assert cnt == o.counters()
        rng = np.random.default_rng(11 + step)
        rays = np.zeros((3000, 8), np.float32); rays[:, 0:3] = rng.uniform([-1.5, 0.1, 0.5], [1.5, 1.5, 4.0], (3000, 3)); d = rng.normal(size=(3000, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 7] = np.inf
        b = osc.intersect_closest(rays)
        for g in (ds.intersect_closest(rays), ds.intersect_stream(rays)):          # the rope copy (query API) and the 8-wide layout (the render kernels' walk)
            for f in ("type", "primitive_id", "geometry_id", "instance_id"): assert np.array_equal(g[f], b[f]), f
            assert np.array_equal(g["distance"].view(np.uint32), b["distance"].view(np.uint32))
        rays[:, 7] = 2.0
        assert np.array_equal(ds.intersect_any(rays), osc.intersect_any(rays))
        r2.close()
    assert costs[1] > costs[0]
    # a transform changing with the vertices: a build; and scene option refit_max_cost_ratio builds by itself once the refits have loosened the tree that far
    ds.update_mesh(big[0], base[big[0]][0], base[big[0]][1]); ds.set_instance_transform(big[0], np.asarray(base[big[0]][2], np.float32)); ds.commit()
    assert ds.refits == 0 and ds.stats.wide_cost == ds.stats.wide_cost_built
    r.close()
    for inst in (0, 1):
        ds2 = mrt.DeviceScene(gpu_ctx, sc0, {"instancing": inst, "refit_max_cost_ratio": 1.5})
        assert ds2.stats.leaf_growth == 1.0
        _, m1 = _deformed_instanced(mrt, (w, h), 0.0005, 0.3); _, m2 = _deformed_instanced(mrt, (w, h), 0.05, 1.1)
        ds2.update_mesh(big[0], m1[big[0]][0], m1[big[0]][1]); ds2.commit()
        s2 = ds2.stats
        assert ds2.refits == 1 and max(s2.wide_cost / s2.wide_cost_built, s2.leaf_growth) <= 1.5 and s2.leaf_growth > 0.9, (inst, s2.wide_cost, s2.wide_cost_built, s2.leaf_growth)
        ds2.update_mesh(big[0], m2[big[0]][0], m2[big[0]][1]); ds2.commit()
        assert ds2.refits == 0 and ds2.stats.wide_cost == ds2.stats.wide_cost_built and ds2.stats.leaf_growth == 1.0, "beyond the ratio: built again"
        ds2.close()


def test_refit_argument_checks_and_fallbacks(mrt, gpu_ctx):
    w, h = 64, 64
    sc = mrt.CornellScene((w, h))
    meshes = mrt.flatten_scene(sc, share=True)
    # a two-level scene refits the changed mesh's BLAS in place (round 6)
    ds = mrt.DeviceScene(gpu_ctx, sc, {"instancing": 1})
    k = max(range(len(meshes)), key=lambda i: len(meshes[i][0]) if meshes[i][4] < 0 else -1)
    pos, nrm = np.asarray(meshes[k][0], np.float32), np.asarray(meshes[k][1], np.float32)
    ds.update_mesh(k, pos * np.float32(1.01), nrm); ds.commit()
    assert ds.refits == 1
    with pytest.raises(mrt.MRTError): ds.update_mesh(k, pos[:-1], nrm[:-1])          # the vertex count must stay
    with pytest.raises(mrt.MRTError): ds.update_mesh(len(meshes) + 3, pos, nrm)
    with pytest.raises(mrt.MRTError): ds.update_mesh(k, pos, nrm[:-1])              # fewer normals than positions: refused by the wrapper before the C side reads past the array
    # scene option refit = 0: the same calls build
    ds2 = mrt.DeviceScene(gpu_ctx, sc, {"refit": 0})
    ds2.update_mesh(k, pos * np.float32(1.01), nrm); ds2.commit()
    assert ds2.refits == 0
    ds3 = mrt.DeviceScene(gpu_ctx, sc)
    ds3.update_mesh(k, pos * np.float32(1.01), nrm); ds3.commit()
    assert ds3.refits == 1


@pytest.mark.parametrize("sopt", [{}, {"refit": 0}, {"instancing": 1}])
def test_commit_while_a_draw_is_in_flight_is_ordered_behind_it(mrt, gpu_ctx, sopt):
    """mrt_renderer_render returns at once; a host that deforms a mesh and commits while those frames are still on the GPU must get them on the OLD geometry and the next draw on the new
    one — the commit's kernels (a refit in place, or a build that frees what the frames read) are ordered behind the draw on the device.  Same image as the sequence with a wait after
    every step; 1080p so that the first draw is still running when the commit arrives."""
    w, h = 1920, 1080
    imgs = []
    for asynchronous in (False, True):
        sc0, base = (_deformed if not sopt.get("instancing") else _deformed_instanced)(mrt, (w, h), 0.0, 0.0)
        _, moved = (_deformed if not sopt.get("instancing") else _deformed_instanced)(mrt, (w, h), 0.03, 0.9)
        big = [k for k, m in enumerate(base) if len(m[0]) > 100000 and m[4] < 0][0]
        r = mrt.Renderer((w, h), sc0, ctx=gpu_ctx, scene_options=sopt); ds = r.device_scene
        r.draw(48, wait=not asynchronous)          # ~25 ms of GPU work: still running when the commit below arrives
        ds.update_mesh(big, moved[big][0], moved[big][1]); ds.commit()
        r.draw(48, wait=not asynchronous)
        ds.update_mesh(big, base[big][0], base[big][1]); ds.commit()
        r.draw(8, wait=True)
        imgs.append((r.accumulation().copy(), r.stats.closest_rays, r.stats.shadow_rays)); r.close()
    assert imgs[0][1:] == imgs[1][1:]
    assert np.array_equal(imgs[0][0].view(np.uint32), imgs[1][0].view(np.uint32))


def test_reads_and_light_updates_while_a_draw_is_in_flight(mrt, gpu_ctx):
    """The other calls a host may make while frames are still on the GPU: reading the accumulation (waits for them), new lights (the frames in flight keep the old ones, the next draw
    gets the new), a new camera (a launch parameter: captured when the draw was issued).  Same images as the sequence with a wait after every step."""
    w, h = 1920, 1080
    out = []
    for asynchronous in (False, True):
        sc = mrt.DragonScene((w, h)); r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
        r.draw(48, wait=not asynchronous)
        a0 = r.accumulation().copy()                       # (reads wait)
        r.draw(24, wait=not asynchronous)
        from metal_raytracing_amd._ffi import _f3
        lights = list(sc.lights); lights[0].color = _f3((9.0, 1.0, 1.0)); r.device_scene.set_lights(lights)
        cam = sc.camera; cam.position = _f3((0.2, 1.1, 5.0)); r.set_camera(cam)
        r.draw(24, wait=not asynchronous)
        a1 = r.accumulation().copy()
        out.append((a0, a1)); r.close()
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert not np.array_equal(out[0][0], out[0][1])
