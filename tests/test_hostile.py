"""The stress stand-in (HostileDragonScene: triangle sizes over 100 : 1, 1 % slivers of up to 50 x their edge) and the builder's answer to it,
triangle pre-splitting (scene option presplit): the image and the ray counts do not depend on it, and the rate on the hostile mesh stays
within a bounded factor of the headline scene's."""
import numpy as np
import pytest


def test_hostile_stand_in_has_the_count_the_extents_and_the_slivers(mrt):
    pos, nrm, subs = mrt.dragon_proxy_hostile()
    idx = subs[0].indices
    assert idx.shape == (871414, 3) and np.isfinite(pos).all() and np.isfinite(nrm).all()
    assert np.allclose(np.abs(pos).max(0), [0.45, 0.317, 0.20], atol=1e-6)
    v = pos[idx]
    ext = (v.max(1) - v.min(1)).max(1)
    assert np.percentile(ext, 99) / np.percentile(ext, 1) >= 100.0                 # 100 : 1 range of triangle sizes
    assert (ext > 20 * np.median(ext)).mean() >= 0.004                            # the slivers
    ref = mrt.dragon_proxy_irregular()[0]
    assert ref.shape[0] < pos.shape[0] <= ref.shape[0] + 8715                     # a sliver owns a duplicated vertex


@pytest.mark.gpu
def test_presplit_changes_the_tree_not_the_image(mrt, orc, gpu_ctx):
    from test_gpu_parity import assert_parity, oracle_render
    w, h = 200, 112
    sc = mrt.HostileDragonScene((w, h))
    imgs, cnts, refs = [], [], []
    for ps in (0, 4, 1.5):
        r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, scene_options={"presplit": ps})
        r.draw(3, wait=True)
        imgs.append(r.accumulation()); cnts.append((r.stats.closest_rays, r.stats.shadow_rays)); refs.append(r.device_scene.stats.bvh_leaves)
        assert r.device_scene.stats.triangles == 885194
        r.close()
    assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[0], imgs[2]) and cnts[0] == cnts[1] == cnts[2]
    ref, cnt = oracle_render(orc, mrt, sc, w, h, 3)
    assert_parity(imgs[1], ref)
    assert cnts[1] == cnt
    # queries through every traversal: brute-force oracle == split tree
    rng = np.random.default_rng(5)
    n = 4000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.uniform(-1, 1, (n, 3)) * [3, 1, 3] + [0, 1.2, 0]
    d = rng.normal(size=(n, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 7] = np.inf
    a = mrt.DeviceScene(gpu_ctx, sc, {"presplit": 0}); b = mrt.DeviceScene(gpu_ctx, sc, {"presplit": 2})
    for f in ("intersect_closest", "intersect_stream"):
        ga, gb = getattr(a, f)(rays), getattr(b, f)(rays)
        for k in ("type", "distance", "instance_id", "geometry_id", "primitive_id", "u", "v"):
            assert np.array_equal(ga[k], gb[k]), (f, k)
    assert np.array_equal(a.intersect_any(rays), b.intersect_any(rays))
    a.close(); b.close()
