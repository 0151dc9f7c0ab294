"""The materials extension (renderer option materials = 1): emission, a specular lobe and dielectric refraction from the
Material fields the reference already carries (ShaderTypes.h:99-107) but never reads (README.md:8 lists them as open work).  The
semantics are defined by the oracle (trace_pixel `if (materials)`); the HIP kernel k_shade<true> must restate them bit for bit, and with
plain diffuse materials the extension must reduce to the reference path exactly."""
import copy

import numpy as np
import pytest

from test_gpu_parity import assert_parity


def _with_material(mrt, model, **fields):
    """a copy of the model's submeshes with edited Material fields (the cached Submesh objects are shared between models)"""
    for mesh in model.meshes:
        subs = []
        for s in mesh.submeshes:
            m = mrt.Material.from_buffer_copy(bytes(s.material))
            for k, v in fields.items():
                if isinstance(v, (list, tuple)):
                    f = getattr(m, k); f.x, f.y, f.z = v
                else:
                    setattr(m, k, v)
            subs.append(mrt.Submesh(s.name, s.indices, m))
        mesh.submeshes = subs
    return model


def _scene(mrt, size, plain=False):
    class S(mrt.CornellScene):
        def __init__(self, size):
            super().__init__(size)
            h = np.pi / 2
            glass = mrt.Model(name="sphere", position=[-0.45, 0.35, 0.35], scale=0.35)
            shiny = mrt.Model(name="sphere", position=[0.45, 0.3, -0.1], scale=0.3)
            lamp = mrt.Model(name="plane", position=[0.99, 1.0, 0.2], rotation=[0, 0, h], scale=0.25)
            if plain:                                             # sphere.mtl carries Ks 0.8 / Ns 32: strip it, so that every lobe choice is the diffuse one
                for mo in (glass, shiny): _with_material(mrt, mo, specular=[0.0, 0.0, 0.0])
            else:
                _with_material(mrt, glass, dissolve=0.15, refractionIndex=1.5, baseColor=[0.9, 0.9, 0.9])
                _with_material(mrt, shiny, specular=[0.8, 0.7, 0.3], specularExponent=96.0, baseColor=[0.2, 0.1, 0.05])
                _with_material(mrt, lamp, emission=[2.0, 1.5, 0.5])
            self.models = self.models[:5] + [glass, shiny, lamp]            # the five walls of CornellScene + three objects
    return S(size)


def test_extension_reduces_to_the_reference_path_for_plain_materials(mrt, orc):
    w, h = 64, 48
    sc = _scene(mrt, (w, h), plain=True)
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    a = orc.OracleRenderer(osc, w, h, max_bounces=4, camera=sc.camera); a.render(2)
    b = orc.OracleRenderer(osc, w, h, max_bounces=4, camera=sc.camera); b.set_materials(True); b.render(2)
    assert np.array_equal(a.accumulation(), b.accumulation()) and a.counters() == b.counters()


def test_oracle_materials_change_the_image_where_expected(mrt, orc):
    w, h = 96, 72
    plain, fancy = _scene(mrt, (w, h), plain=True), _scene(mrt, (w, h))
    op = orc.OracleScene(mrt.flatten_scene(plain), plain.lights); of = orc.OracleScene(mrt.flatten_scene(fancy), fancy.lights)
    a = orc.OracleRenderer(op, w, h, max_bounces=4, camera=plain.camera); a.render(8)
    b = orc.OracleRenderer(of, w, h, max_bounces=4, camera=fancy.camera); b.set_materials(True); b.render(8)
    ia, ib = a.accumulation()[..., :3], b.accumulation()[..., :3]
    assert np.isfinite(ib).all() and (ib >= 0).all()
    assert np.abs(ia - ib).max() > 0.2                                  # the lamp is visible, the spheres look different
    # fewer shadow rays: specular and refracted bounces cast none
    assert b.counters()[1] < a.counters()[1]
    # without the option the fields are ignored, as in the reference kernel
    c = orc.OracleRenderer(of, w, h, max_bounces=4, camera=fancy.camera); c.render(8)
    d = orc.OracleRenderer(op, w, h, max_bounces=4, camera=plain.camera); d.render(8)
    assert not np.array_equal(c.accumulation(), b.accumulation())
    lit = np.abs(c.accumulation()[..., :3] - d.accumulation()[..., :3]).max()
    assert lit > 0                                                      # base colours differ, so the diffuse-only images differ too


@pytest.mark.gpu
def test_gpu_materials_match_the_oracle_bit_for_bit(mrt, orc, gpu_ctx):
    w, h = 160, 120
    sc = _scene(mrt, (w, h))
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx, max_bounces=4)
    r.set_option("materials", 1)
    r.draw(6, wait=True)
    ref = orc.OracleRenderer(osc, w, h, max_bounces=4, camera=sc.camera); ref.set_materials(True); ref.render(6)
    assert_parity(r.accumulation(), ref.accumulation())
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    # option off again: the reference kernel, same renderer
    r.set_option("materials", 0); r.frameIndex = 0; r.reset_stats()
    r.draw(2, wait=True)
    ref0 = orc.OracleRenderer(osc, w, h, max_bounces=4, camera=sc.camera); ref0.render(2)
    assert_parity(r.accumulation(), ref0.accumulation())
    with pytest.raises(mrt.MRTError):
        r.set_option("max_bounces", 17); r.set_option("materials", 1)
    r.close()


@pytest.mark.gpu
def test_gpu_materials_on_the_benchmark_scene(mrt, orc, gpu_ctx):
    """DragonScene's own MTL values (Ks 0.2 / 0.8, Ns 37 ... 155 on the train, the dragon and the spheres) through the extension."""
    w, h = 192, 108
    sc = mrt.DragonScene((w, h))
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx)
    r.set_option("materials", 1)
    r.draw(3, wait=True)
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    ref = orc.OracleRenderer(osc, w, h, camera=sc.camera); ref.set_materials(True); ref.render(3)
    assert_parity(r.accumulation(), ref.accumulation())
    assert (r.stats.closest_rays, r.stats.shadow_rays) == ref.counters()
    r.close()
