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


def test_refit_argument_checks_and_fallbacks(mrt, gpu_ctx):
    w, h = 64, 64
    sc = mrt.CornellScene((w, h))
    meshes = mrt.flatten_scene(sc, share=True)
    # a two-level scene builds again (the BLAS refit is not implemented): the image still follows the vertices
    ds = mrt.DeviceScene(gpu_ctx, sc, {"instancing": 1})
    k = max(range(len(meshes)), key=lambda i: len(meshes[i][0]) if meshes[i][4] < 0 else -1)
    pos, nrm = np.asarray(meshes[k][0], np.float32), np.asarray(meshes[k][1], np.float32)
    ds.update_mesh(k, pos * np.float32(1.01), nrm); ds.commit()
    assert ds.refits == 0
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
