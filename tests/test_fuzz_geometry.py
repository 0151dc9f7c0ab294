"""Hostile geometry against the oracle, flattened and two-level: pole fans (hundreds of slivers around one vertex — the drain phase's cooperative
triangle tests), exactly coincident triangles (closest-hit ties resolved by the lowest id), zero-area and needle triangles, sizes over four decades,
stacked coplanar sheets; queries through both traversals and a 4-bounce image, bit for bit."""
import numpy as np
import pytest

from test_gpu_parity import assert_parity


def _material(mrt, rgb):
    m = mrt.Material()
    m.baseColor = mrt.Float3(*rgb); m.dissolve = 1.0
    return m


def _fan(rng, n):
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rim = np.c_[np.cos(ang), 0.05 * rng.normal(size=n), np.sin(ang)] * rng.uniform(0.6, 1.0, (n, 1))
    pos = np.vstack([[0.0, 0.3, 0.0], rim]).astype(np.float32)
    idx = np.array([[0, 1 + k, 1 + (k + 1) % n] for k in range(n)], np.uint32)
    return pos, idx


def _soup(rng, n):
    c = rng.uniform(-1, 1, (n, 3)) * [1.5, 0.8, 1.5] + [0, 1.0, 0]
    size = 10.0 ** rng.uniform(-3.5, -0.3, (n, 1, 1))
    tri = c[:, None, :] + rng.normal(size=(n, 3, 3)) * size
    tri[::17, 2] = tri[::17, 1]                          # zero-area: two equal vertices
    tri[5::23, 2] = 0.5 * (tri[5::23, 0] + tri[5::23, 1])   # zero-area: collinear
    tri[3::11] = tri[2::11][: len(tri[3::11])]           # exact duplicates of the previous triangle: a tie on t for every ray that hits them
    pos = tri.reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    return pos, idx


def _sheets(k):
    g = np.linspace(-1, 1, k + 1)
    xs, zs = np.meshgrid(g, g)
    pos = np.c_[xs.ravel(), np.zeros(xs.size), zs.ravel()].astype(np.float32)
    q = np.array([[r * (k + 1) + c, r * (k + 1) + c + 1, (r + 1) * (k + 1) + c + 1, (r + 1) * (k + 1) + c] for r in range(k) for c in range(k)], np.uint32)
    idx = np.vstack([q[:, [0, 1, 2]], q[:, [0, 2, 3]], q[:, [0, 1, 2]]])     # the first half of every quad twice: coincident triangles in one mesh
    return pos, idx


def _normals(pos, idx):
    n = np.zeros_like(pos)
    fn = np.cross(pos[idx[:, 1]] - pos[idx[:, 0]], pos[idx[:, 2]] - pos[idx[:, 0]])
    for k in range(3): np.add.at(n, idx[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return np.where(ln > 1e-20, n / np.maximum(ln, 1e-20), [0.0, 1.0, 0.0]).astype(np.float32)


class _Raw:
    """stands in for Model: one mesh from arrays"""
    def __init__(self, mrt, name, pos, idx, rgb, position, rotation, scale, share=None):
        self.name = name
        if share is not None:
            src = share.meshes[0]
            self.meshes = [mrt.Mesh(name, src.positions, src.normals, src.submeshes, position, rotation, scale)]
        else:
            self.meshes = [mrt.Mesh(name, pos, _normals(pos, idx), [mrt.Submesh(name, idx, _material(mrt, rgb))], position, rotation, scale)]


def _scene(mrt, size, seed):
    rng = np.random.default_rng(seed)
    class S(mrt.Scene):
        def __init__(self, size):
            super().__init__(size)
            fan = _Raw(mrt, "fan", *_fan(rng, 300), (0.8, 0.3, 0.2), [0.2, 0.4, 0.6], [0.3, 0.1, -0.2], 0.9)
            self.models = [mrt.Model(name="plane", position=[0, 0, 0], scale=10),
                           fan,
                           _Raw(mrt, "fan", None, None, None, [-1.1, 0.9, 0.2], [1.2, 0.4, 0.3], 0.5, share=fan),
                           _Raw(mrt, "fan", None, None, None, [0.9, 1.3, 1.1], [-0.7, 2.0, 0.1], 0.35, share=fan),
                           _Raw(mrt, "soup", *_soup(rng, 1500), (0.3, 0.7, 0.4), [0, 0, 0.3], [0, 0, 0], 1.0),
                           _Raw(mrt, "sheets", *_sheets(12), (0.5, 0.5, 0.9), [0.1, 0.55, 1.0], [0.05, 0.3, 0.0], 0.8)]
    return S(size)


def _rays(rng, n):
    o = np.array([0.0, 1.2, 4.0]) + rng.normal(size=(n, 3)) * 0.5
    t = np.c_[rng.uniform(-1.8, 1.8, n), rng.uniform(0, 1.8, n), rng.uniform(-1.5, 2.0, n)]
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32); r[:, 0:3] = o; r[:, 4:7] = d; r[:, 7] = np.inf
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("two_level", [False, True])
def test_hostile_geometry(mrt, orc, gpu_ctx, seed, two_level):
    w, h = 96, 64
    sc = _scene(mrt, (w, h), seed)
    osc = orc.OracleScene(mrt.flatten_scene(sc, share=two_level), sc.lights, instancing=two_level)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=4, scene_options={"instancing": 1} if two_level else None)
    rays = _rays(np.random.default_rng(100 + seed), 20000)
    o = osc.intersect_closest(rays)
    assert (o["type"] == 1).mean() > 0.6
    for g in (r.device_scene.intersect_closest(rays), r.device_scene.intersect_stream(rays)):
        for f in ("type", "distance", "instance_id", "geometry_id", "primitive_id", "u", "v"):
            assert np.array_equal(g[f], o[f]), f
    rays[:, 7] = 2.5
    oa = osc.intersect_any(rays)
    assert np.array_equal(r.device_scene.intersect_any(rays), oa)
    assert np.array_equal(r.device_scene.intersect_stream(rays, any_hit=True)["type"], oa)
    r.draw(3, wait=True)
    ref = orc.OracleRenderer(osc, w, h, max_bounces=4, camera=sc.camera); ref.render(3)
    assert_parity(r.accumulation(), ref.accumulation(), exact_frac=1.0)
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    r.close()
