"""CPU-side checks of the product: the C-ABI library loads and exports every symbol
include/mrt_abi.h declares, the data contract has the reference's layout (ShaderTypes.h:60-107),
and the host geometry helpers (OBJ/MTL reader, T*R*S, camera, procedural meshes) agree with
independent restatements.  No compute calls: there is no GPU here."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "assets", "Resources")


def test_library_exports_every_declared_symbol(mrt):
    # the host contract (mrt_abi.h) and the diagnostics header beside it (mrt_debug.h: tests, tools, bench.py — not installed with the ABI header)
    hdr = open(os.path.join(ROOT, "include", "mrt_abi.h")).read() + open(os.path.join(ROOT, "include", "mrt_debug.h")).read()
    hdr = re.sub(r"#ifdef MRT_DIAGNOSTICS.*?#endif", "", hdr, flags=re.S)          # (what only a diagnostics build exports)
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mrt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    abi_only = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mrt_abi.h")).read(), flags=re.S)
    assert not re.findall(r"\bmrt_debug_[a-z0-9_]+\s*\(", abi_only), "mrt_debug_* entry points belong in mrt_debug.h"
    raw = C.CDLL(mrt.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in include/ but not exported"
    from metal_raytracing_amd import _ffi
    assert declared == set(_ffi.SIGNATURES), "ctypes table and headers disagree"
    assert not hasattr(raw, "mrt_debug_poke_wnode"), "the release library must not export the node-poking aid (MRT_DIAGNOSTICS builds only)"
    assert mrt.lib.mrt_abi_version() == 3


def test_struct_layout_matches_shader_types(mrt):
    L, M, U, Cm = mrt.Light, mrt.Material, mrt.Uniforms, mrt.Camera
    assert C.sizeof(Cm) == 64 and C.sizeof(L) == 128 and C.sizeof(U) == 96 and C.sizeof(M) == 64
    assert [getattr(L, f).offset for f in ("type", "position", "color", "forward", "right", "up", "coneAngle", "direction")] == [0, 16, 32, 48, 64, 80, 96, 112]
    assert [getattr(U, f).offset for f in ("width", "height", "blocksWide", "frameIndex", "lightCount", "camera")] == [0, 4, 8, 12, 16, 32]
    assert [getattr(M, f).offset for f in ("baseColor", "specular", "emission", "specularExponent", "refractionIndex", "dissolve")] == [0, 16, 32, 48, 52, 56]
    assert [getattr(Cm, f).offset for f in ("position", "right", "up", "forward")] == [0, 16, 32, 48]
    assert mrt.LightType.sunlight == 1 and mrt.LightType.spotlight == 2 and mrt.LightType.pointlight == 3 and mrt.LightType.areaLight == 4


def test_no_gpu_means_loud_failure_not_fallback(mrt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mrt.MRTError) as e:
        mrt.Context(0)
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "metal-raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "mrt_oracle" not in txt and "import oracle" not in txt and "oracle/" not in txt.replace("nothing under oracle/", ""), f


# ---------------------------------------------------------------- OBJ/MTL reader vs an independent parser
def _py_obj(path):
    V, N, faces, mats, cur = [], [], [], [], None
    for line in open(path):
        t = line.split()
        if not t: continue
        if t[0] == "v": V.append([float(x) for x in t[1:4]])
        elif t[0] == "vn": N.append([float(x) for x in t[1:4]])
        elif t[0] == "usemtl": cur = t[1]; mats.append([cur, 0])
        elif t[0] == "f":
            if not mats: mats.append(["", 0])
            vs = []
            for c in t[1:]:
                p = c.split("/")
                vi = int(p[0]); ni = int(p[2]) if len(p) > 2 and p[2] else 0
                vs.append((vi - 1 if vi > 0 else len(V) + vi, ni - 1 if ni > 0 else (len(N) + ni if ni < 0 else -1)))
            for k in range(1, len(vs) - 1):
                faces.append((vs[0], vs[k], vs[k + 1])); mats[-1][1] += 1
    return np.array(V, np.float32), np.array(N, np.float32), faces, [m for m in mats if m[1]]


@pytest.mark.parametrize("name,tris", [("plane", 2), ("plane-back", 2), ("sphere", 4900), ("train", 3624), ("treefir", 352), ("teapot", 15704)])
def test_obj_reader(mrt, name, tris):
    pos, nrm, subs = mrt.load_obj(os.path.join(RES, name + ".obj"))
    V, N, faces, mats = _py_obj(os.path.join(RES, name + ".obj"))
    assert sum(s.triangleCount for s in subs) == tris == len(faces)                 # counts of SURVEY §8 a-12
    assert [s.name for s in subs] == [m[0] for m in mats] and [s.triangleCount for s in subs] == [m[1] for m in mats]
    idx = np.concatenate([s.indices for s in subs])
    assert idx.max() < len(pos)
    assert len(pos) == len({(a, b) for f in faces for (a, b) in f})                 # one vertex per distinct (v,vn)
    for k in np.random.default_rng(0).integers(0, len(faces), 200):
        for c in range(3):
            vi, ni = faces[k][c]
            assert np.array_equal(pos[idx[k, c]], V[vi])
            if ni >= 0: assert np.array_equal(nrm[idx[k, c]], N[ni])
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=2e-3)


def test_mtl_reader(mrt):
    _, _, subs = mrt.load_obj(os.path.join(RES, "train.obj"))
    kd = {s.name: s.material.baseColor.tolist() for s in subs}
    assert np.allclose(kd["Body"], [0.913099, 0.597202, 0.059511]) and np.allclose(kd["Hub"], [0.8, 0.8, 0.8])
    m = [s for s in subs if s.name == "Body"][0].material
    assert abs(m.specularExponent - 37.254902) < 1e-5 and m.refractionIndex == 1.0 and m.dissolve == 1.0 and np.allclose(m.specular.tolist(), [0.2] * 3)
    _, _, subs = mrt.load_obj(os.path.join(RES, "plane.obj"))
    assert np.allclose(subs[0].material.baseColor.tolist(), [0.5, 0.5, 0.5])        # plane.mtl:7
    _, nrm, subs = mrt.load_obj(os.path.join(RES, "teapot.obj"))                    # default.mtl missing, no vn
    assert np.allclose(subs[0].material.baseColor.tolist(), [0.8, 0.8, 0.8]) and len(subs) == 2


def test_obj_errors(mrt, tmp_path):
    with pytest.raises(mrt.MRTError) as e:
        mrt.load_obj(str(tmp_path / "missing.obj"))
    assert e.value.code == 4
    p = tmp_path / "bad.obj"; p.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(mrt.MRTError):
        mrt.load_obj(str(p))
    p = tmp_path / "empty.obj"; p.write_text("v 0 0 0\n")
    with pytest.raises(mrt.MRTError):
        mrt.load_obj(str(p))
    p = tmp_path / "neg.obj"; p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf -4 -3 -1 -2\n")
    pos, nrm, subs = mrt.load_obj(str(p))
    assert subs[0].triangleCount == 2 and np.allclose(np.abs(nrm[:, 2]), 1)


def test_procedural_meshes(mrt):
    pos, nrm, subs = mrt.dragon_proxy()
    assert subs[0].triangleCount == 871414 and len(subs) == 1                        # Stanford dragon count
    assert np.allclose(pos.max(0), [0.45, 0.317, 0.20], atol=1e-6) and np.allclose(pos.min(0), [-0.45, -0.317, -0.20], atol=1e-6)
    assert subs[0].material.baseColor.tolist() == [1.0, 0.0, 0.0]                    # dragon.mtl:6
    assert subs[0].indices.max() == len(pos) - 1 and np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-3)
    tri = pos[subs[0].indices[::997]]
    area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    assert (area > 0).mean() > 0.999
    pos2, _, subs2 = mrt.dragon_proxy()
    assert np.array_equal(pos, pos2) and np.array_equal(subs[0].indices, subs2[0].indices)   # deterministic
    _, _, b = mrt.bunny_proxy()
    assert b[0].triangleCount == 69451


def test_transform_and_camera_match_oracle(mrt, orc):
    rng = np.random.default_rng(1)
    for _ in range(20):
        p, r, s = rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 3), float(rng.uniform(0.1, 5))
        assert np.array_equal(mrt.make_transform(p, r, s), orc.make_transform(p, r, s))
    for (w, h) in [(1920, 1080), (256, 256), (800, 600), (3840, 2160), (7, 5)]:
        a, b = mrt.Scene.setupCamera((w, h)), orc.default_camera(w, h)
        assert bytes(a) == bytes(b)


def test_dragon_scene_definition(mrt):
    sc = mrt.DragonScene((1920, 1080))                                              # DragonScene.swift:14-22
    assert [m.name for m in sc.models] == ["train", "dragon", "treefir", "plane", "sphere", "sphere", "plane-back"]
    assert [m.triangleCount for m in sc.meshes] == [3624, 871414, 352, 2, 4900, 4900, 2] and sc.triangleCount == 885194
    assert max(len(m.submeshes) for m in sc.meshes) == 6                            # maxSubmeshes (Renderer.swift:128)
    assert len(sc.lights) == 2 and sc.lights[0].type == 4 and sc.lights[1].type == 2  # Scene.swift:21-30
    assert abs(sc.lights[1].coneAngle - 25 / 180 * math.pi) < 1e-6
    g = np.load(os.path.join(ROOT, "tests", "golden", "dragonscene_setup.npz"))
    assert np.array_equal(np.stack([m.transform for m in sc.meshes]), g["transforms"])


def test_layout_limits_reject_scenes_the_traversal_cannot_address(mrt):
    """mrt_scene_commit's size check (bvh_build.hip layout_limits): the rope layout packs a child index into 24 bits and
    addresses nodes + packets through one 32-bit byte offset, so a scene beyond either must be refused, not traversed."""
    L = mrt.lib.mrt_debug_layout_limits
    OK, UNSUPPORTED = 0, mrt._ffi.MRT_ERR_UNSUPPORTED
    assert L(0, 0) == OK and L(1, 1) == OK and L(885194, 137314) == OK
    assert L(1 << 23, 0) == OK                                   # 2n-1 < 2^24 nodes, 176 n < 2^32
    assert L(24_000_000, (1 << 24) - 1) == OK                    # fits the byte offset; node count is what the collapse leaves
    assert L(24_000_000, 1 << 24) == UNSUPPORTED                 # one node too many for the 24-bit child index
    assert L(24_500_000, 0) == UNSUPPORTED                       # 64 (2n-1) + 48 n > 2^32
    assert L(1 << 26, 0) == UNSUPPORTED
    assert b"too large" in mrt.lib.mrt_last_error()


def _sah_cost(lo, hi, order, left, right, parent):
    """SAH of a binary tree over boxes (Ct = Ci = 1, one reference per leaf), iteratively bottom-up."""
    n = len(order)
    nlo = np.empty((2 * n - 1, 3), np.float64); nhi = np.empty((2 * n - 1, 3), np.float64)
    nlo[n - 1:] = lo[order]; nhi[n - 1:] = hi[order]
    depth = np.zeros(2 * n - 1, np.int64)
    root = int(np.where(parent == 0xFFFFFFFF)[0][0])
    # parents before children in a BFS from the root
    orderq = [root]
    for q in orderq:
        if q < n - 1:
            for c in (int(left[q]), int(right[q])):
                depth[c] = depth[q] + 1; orderq.append(c)
    assert len(orderq) == 2 * n - 1
    for q in reversed(orderq):
        if q < n - 1:
            l, r = int(left[q]), int(right[q])
            nlo[q] = np.minimum(nlo[l], nlo[r]); nhi[q] = np.maximum(nhi[l], nhi[r])
    d = nhi - nlo
    area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
    return float(area.sum() / area[root]), int(depth.max())


def test_host_sah_builder_makes_a_valid_tree_of_lower_cost_than_a_median_split(mrt):
    """Scene option builder = 2: the host part (binned SAH over reference boxes) — every reference is a leaf exactly once, every internal node
    has two children whose parent it is, one root; and its SAH cost beats an object-median tree over the same boxes."""
    import ctypes as C
    rng = np.random.default_rng(11)
    n = 20000
    c = np.concatenate([rng.normal(0, 1, (n // 2, 3)), rng.normal(0, 0.05, (n // 2, 3)) + [3, 0, 0]]).astype(np.float32)      # two clusters of very different density
    e = (rng.uniform(0.001, 0.02, (n, 3)) * np.where(rng.uniform(size=(n, 1)) < 0.01, 30, 1)).astype(np.float32)            # 1 % long boxes
    lo4 = np.zeros((n, 4), np.float32); hi4 = np.zeros((n, 4), np.float32); lo4[:, :3] = c - e; hi4[:, :3] = c + e
    order = np.zeros(n, np.uint32); left = np.zeros(n - 1, np.uint32); right = np.zeros(n - 1, np.uint32); parent = np.zeros(2 * n - 1, np.uint32)
    mrt._ffi.check(mrt.lib.mrt_debug_host_sah(mrt._ffi.ptr(lo4), mrt._ffi.ptr(hi4), n, mrt._ffi.ptr(order), mrt._ffi.ptr(left), mrt._ffi.ptr(right), mrt._ffi.ptr(parent)))
    assert np.array_equal(np.sort(order), np.arange(n))                                   # a permutation
    assert (parent == 0xFFFFFFFF).sum() == 1
    kids = np.concatenate([left, right])
    assert np.array_equal(np.sort(kids), np.setdiff1d(np.arange(2 * n - 1), np.where(parent == 0xFFFFFFFF)[0]))      # every non-root node is a child exactly once
    assert np.array_equal(parent[left], np.arange(n - 1)) and np.array_equal(parent[right], np.arange(n - 1))
    # the numbering does not depend on the threads' timing: a second build gives the same arrays
    o2 = np.zeros_like(order); l2 = np.zeros_like(left); r2 = np.zeros_like(right); p2 = np.zeros_like(parent)
    mrt._ffi.check(mrt.lib.mrt_debug_host_sah(mrt._ffi.ptr(lo4), mrt._ffi.ptr(hi4), n, mrt._ffi.ptr(o2), mrt._ffi.ptr(l2), mrt._ffi.ptr(r2), mrt._ffi.ptr(p2)))
    assert np.array_equal(order, o2) and np.array_equal(left, l2) and np.array_equal(right, r2) and np.array_equal(parent, p2)
    cost, depth = _sah_cost(lo4[:, :3].astype(np.float64), hi4[:, :3].astype(np.float64), order, left, right, parent)
    # object-median tree over the same boxes
    m_order = np.zeros(n, np.uint32); m_left = np.zeros(n - 1, np.uint32); m_right = np.zeros(n - 1, np.uint32); m_parent = np.full(2 * n - 1, 0xFFFFFFFF, np.uint32)
    cen = lo4[:, :3] + hi4[:, :3]
    ids = np.arange(n); nxt = [0]
    def build(b, e_, par):
        if e_ - b == 1:
            m_parent[n - 1 + b] = par; return n - 1 + b
        me = nxt[0]; nxt[0] += 1; m_parent[me] = par
        sub = ids[b:e_]; ax = int(np.argmax(cen[sub].max(0) - cen[sub].min(0)))
        ids[b:e_] = sub[np.argsort(cen[sub, ax], kind="stable")]
        mid = (b + e_) // 2
        m_left[me] = build(b, mid, me); m_right[me] = build(mid, e_, me)
        return me
    import sys
    sys.setrecursionlimit(10000)
    build(0, n, 0xFFFFFFFF)
    m_order[:] = ids
    mcost, _ = _sah_cost(lo4[:, :3].astype(np.float64), hi4[:, :3].astype(np.float64), m_order, m_left, m_right, m_parent)
    assert cost < 0.9 * mcost, (cost, mcost)
    assert depth < 64


def test_image_writers_round_trip(mrt, tmp_path):
    """SURVEY f-1: the tonemapped image as PNG (zlib only — read back here with an independent decoder when there is one, by hand otherwise) and the radiance buffer as PFM
    (rows bottom to top, as the accumulation buffer lies: Raytracing.metal has no y-flip, SURVEY a-4)."""
    import struct, zlib
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    p = str(tmp_path / "a.png"); mrt.save_png(p, img)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8]); data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        if tag == b"IHDR": ihdr = struct.unpack(">IIBBBBB", data)
        if tag == b"IDAT": idat += data
        pos += 12 + n
    assert ihdr == (53, 37, 8, 6, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(37, 1 + 53 * 4)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(37, 53, 4), img)
    try:
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(p)), img)
    except ImportError:
        pass
    acc = rng.random((11, 7, 4), dtype=np.float32) * 3.0
    q = str(tmp_path / "a.pfm"); mrt.save_pfm(q, acc)
    b = open(q, "rb").read()
    head = b"PF\n7 11\n-1.0\n"
    assert b.startswith(head) and np.array_equal(np.frombuffer(b[len(head):], "<f4").reshape(11, 7, 3), acc[:, :, :3])
    with pytest.raises(ValueError): mrt.save_png(p, img[:, :, :3])
