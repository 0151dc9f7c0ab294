"""Host-side mirror of the reference's scene graph — Scene, DragonScene, Model, Mesh, Submesh
(Scene.swift, DragonScene.swift, Model.swift, Mesh.swift, SubMesh.swift) — over the C ABI.

Same names, initialiser arguments and properties as the Swift classes, so the parity tests read
like tests of the reference would.  Geometry ingest (OBJ/MTL, T*R*S, camera) is done by the
library's host helpers (mrt_obj_load, mrt_make_transform, mrt_default_camera); this file only
holds numpy views of the results and hands them to the device side through mrt_scene_add_mesh /
mrt_mesh_add_submesh.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _ffi
from ._ffi import Camera, Float3, Light, Material, check, lib, ptr

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resource_dirs():
    dirs = []
    if os.environ.get("MRT_RESOURCES"):
        dirs.append(os.environ["MRT_RESOURCES"])
    dirs.append(os.path.join(_REPO, "assets", "Resources"))
    return dirs


def find_resource(name):
    """Bundle.main.url(forResource: "Resources/<name>", withExtension: "obj") (Model.swift:14)."""
    for d in resource_dirs():
        p = os.path.join(d, name + ".obj")
        if os.path.exists(p):
            return p
    return None


class Submesh:
    """SubMesh.swift:10-33 — one index buffer + one Material."""

    def __init__(self, name, indices, material):
        self.name = name
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        self.material = material

    @property
    def triangleCount(self):
        return int(self.indices.shape[0])


def make_transform(position, rotation, scale):
    """translate * rotate(Rx*Ry*Rz) * scale (Mesh.swift:21-24, Utilities.swift:104-166); (4,4) [col][row]."""
    p = (C.c_float * 3)(*map(float, position))
    r = (C.c_float * 3)(*map(float, rotation))
    out = (C.c_float * 16)()
    check(lib.mrt_make_transform(p, r, float(scale), out))
    return np.array(out, dtype=np.float32).reshape(4, 4)


class Mesh:
    """Mesh.swift:10-49 — vertex/normal buffers, transform, submeshes."""

    def __init__(self, modelName, positions, normals, submeshes, position, rotation, scale):
        self.modelName = modelName
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.transform = make_transform(position, rotation, scale)
        self.submeshes = list(submeshes)

    @property
    def triangleCount(self):
        return sum(s.triangleCount for s in self.submeshes)


def _meshdata_to_python(handle, name):
    nverts, nsub = C.c_size_t(), C.c_int32()
    check(lib.mrt_meshdata_counts(handle, C.byref(nverts), C.byref(nsub)))
    pos = np.empty((nverts.value, 3), np.float32)
    nrm = np.empty((nverts.value, 3), np.float32)
    check(lib.mrt_meshdata_vertices(handle, ptr(pos), ptr(nrm)))
    subs = []
    for s in range(nsub.value):
        nt = C.c_size_t()
        check(lib.mrt_meshdata_submesh(handle, s, C.byref(nt), None, None, None, 0))
        idx = np.empty((nt.value, 3), np.uint32)
        mat = Material()
        buf = C.create_string_buffer(256)
        check(lib.mrt_meshdata_submesh(handle, s, C.byref(nt), ptr(idx), C.byref(mat), buf, 256))
        subs.append(Submesh(buf.value.decode(), idx, mat))
    return pos, nrm, subs


def load_obj(path):
    h = C.c_void_p()
    check(lib.mrt_obj_load(path.encode(), C.byref(h)))
    try:
        return _meshdata_to_python(h, os.path.basename(path))
    finally:
        lib.mrt_meshdata_free(h)


def dragon_proxy():
    h = C.c_void_p()
    check(lib.mrt_dragon_proxy(C.byref(h)))
    try:
        return _meshdata_to_python(h, "dragon")
    finally:
        lib.mrt_meshdata_free(h)


def dragon_proxy_irregular():
    h = C.c_void_p()
    check(lib.mrt_dragon_proxy_irregular(C.byref(h)))
    try:
        return _meshdata_to_python(h, "dragon-irregular")
    finally:
        lib.mrt_meshdata_free(h)


def dragon_proxy_hostile():
    h = C.c_void_p()
    check(lib.mrt_dragon_proxy_hostile(C.byref(h)))
    try:
        return _meshdata_to_python(h, "dragon-hostile")
    finally:
        lib.mrt_meshdata_free(h)


def bunny_proxy():
    h = C.c_void_p()
    check(lib.mrt_bunny_proxy(C.byref(h)))
    try:
        return _meshdata_to_python(h, "bunny")
    finally:
        lib.mrt_meshdata_free(h)


_PROXIES = {"dragon": dragon_proxy, "dragon-irregular": dragon_proxy_irregular, "dragon-hostile": dragon_proxy_hostile, "bunny": bunny_proxy}
_mesh_cache = {}


class Model:
    """Model.swift:10-40 — Model(name:position:rotation:scale:).  `on device` has no counterpart:
    the device is chosen when the Renderer is created."""

    def __init__(self, name, position, rotation=(0.0, 0.0, 0.0), scale=1.0):
        self.name = name
        self.position, self.rotation, self.scale = tuple(position), tuple(rotation), float(scale)
        path = find_resource(name)
        self.source = path if path else None
        key = path or ("proxy:" + name)
        if key not in _mesh_cache:
            if path:
                _mesh_cache[key] = load_obj(path)
            elif name in _PROXIES:
                _mesh_cache[key] = _PROXIES[name]()
            else:
                raise FileNotFoundError(f"Resources/{name}.obj not found in {resource_dirs()}")
        pos, nrm, subs = _mesh_cache[key]
        if not path:
            self.source = f"procedural proxy ({sum(s.triangleCount for s in subs)} triangles)"
        self.meshes = [Mesh(name, pos, nrm, subs, position, rotation, scale)]


class Scene:
    """Scene.swift:10-67."""

    def __init__(self, size):
        self.size = (int(size[0]), int(size[1]))
        self.camera = Scene.setupCamera(self.size)
        self.models = []
        light1 = Scene.setupLight()
        light3 = Light.spotLight(position=[2, 1, 4], direction=[-1.5, -0.5, -1.5], coneAngle=np.float32(25.0) / np.float32(180.0) * np.float32(math.pi), color=[4, 4, 4])
        self.lights = [light1, light3]                       # Scene.swift:30

    def updateUniforms(self, size):                          # Scene.swift:36-38
        self.size = (int(size[0]), int(size[1]))
        self.camera = Scene.setupCamera(self.size)

    @staticmethod
    def setupCamera(size):                                   # Scene.swift:40-57
        cam = Camera()
        check(lib.mrt_default_camera(int(size[0]), int(size[1]), C.byref(cam)))
        return cam

    @staticmethod
    def setupLight():                                        # Scene.swift:59-67
        return Light.areaLight(position=[0.0, 1.98, 0.0], forward=[0.0, -1.0, 0.0], right=[0.25, 0.0, 0.0], up=[0.0, 0.0, 0.25], color=[4.0, 4.0, 4.0])

    @property
    def meshes(self):
        return [m for model in self.models for m in model.meshes]   # scene.models.flatMap(\.meshes)

    @property
    def triangleCount(self):
        return sum(m.triangleCount for m in self.meshes)

    def describe(self):
        return {m.name: m.source for m in self.models}


class DragonScene(Scene):
    """DragonScene.swift:10-34 — the benchmark scene (BASELINE.json configs[1])."""

    def __init__(self, size):
        super().__init__(size)
        pi = math.pi
        self.models = [
            Model(name="train", position=[-0.3, 0, 0.4], scale=0.5),
            Model(name="dragon", position=[0.3, 0.38, 2.5], rotation=[0, np.float32(np.float32(pi) / np.float32(2) * np.float32(1.2)), 0], scale=1.2),
            Model(name="treefir", position=[0.5, 0, -0.2], scale=0.7),
            Model(name="plane", position=[0, 0, 0], scale=10),
            Model(name="sphere", position=[-1.9, 0.0, 0.3], scale=1),
            Model(name="sphere", position=[2.9, 0.0, -0.5], scale=2),
            Model(name="plane-back", position=[0, 0, -1.5], scale=10),
        ]


class IrregularDragonScene(DragonScene):
    """DragonScene with the second dragon stand-in (irregular connectivity, uneven triangle sizes, shuffled order): the headline
    number must not be a property of the regular tube grid of the first stand-in (assets/README.md)."""

    def __init__(self, size):
        super().__init__(size)
        d = self.models[1]
        assert d.name == "dragon"
        self.models[1] = Model(name="dragon-irregular", position=d.position, rotation=d.rotation, scale=d.scale)


class HostileDragonScene(DragonScene):
    """DragonScene with the stress stand-in: triangle sizes over a 100 : 1 range and 1 % of the triangles pulled into slivers of up to 50 times
    their edge.  A scanned dragon is closer to this than to a uniform tube grid; the rate on it is bounded by a test (tests/test_hostile.py)."""

    def __init__(self, size):
        super().__init__(size)
        d = self.models[1]
        assert d.name == "dragon"
        self.models[1] = Model(name="dragon-hostile", position=d.position, rotation=d.rotation, scale=d.scale)


class CornellScene(Scene):
    """BASELINE.json configs[0]: plane.obj x5 + sphere.obj, one area light.  Not a
    reference scene — built from reference assets for the CPU-runnable plumbing case."""

    def __init__(self, size):
        super().__init__(size)
        h = math.pi / 2
        self.models = [
            Model(name="plane", position=[0, 0, 0], scale=1),
            Model(name="plane", position=[0, 2, 0], rotation=[math.pi, 0, 0], scale=1),
            Model(name="plane", position=[0, 1, -1], rotation=[h, 0, 0], scale=1),
            Model(name="plane", position=[-1, 1, 0], rotation=[0, 0, -h], scale=1),
            Model(name="plane", position=[1, 1, 0], rotation=[0, 0, h], scale=1),
            Model(name="sphere", position=[0, 0.5, 0], scale=0.5),
        ]
        self.lights = [Scene.setupLight()]


class InstancedDragonScene(DragonScene):
    """BASELINE.json configs[4]: dragon at 4x instancing (~3.49 M triangles)."""

    def __init__(self, size, copies=4):
        super().__init__(size)
        pi = math.pi
        extra = [([-1.2, 0.38, 1.6], 0.4), ([1.5, 0.38, 1.2], 2.1), ([-0.2, 0.38, -0.9], 3.3)]
        for k in range(copies - 1):
            pos, ry = extra[k % len(extra)]
            pos = [pos[0] + 0.9 * (k // len(extra)), pos[1], pos[2]]
            self.models.append(Model(name="dragon", position=pos, rotation=[0, ry, 0], scale=1.2))


class GardenScene(Scene):
    """BASELINE.json configs[3]: bunny + teapot + treefir multi-instance, spot + sun lights."""

    def __init__(self, size):
        super().__init__(size)
        self.models = [Model(name="plane", position=[0, 0, 0], scale=10), Model(name="plane-back", position=[0, 0, -2.5], scale=10)]
        k = 0
        for z in (-1.2, 0.4, 2.0):
            for x in (-2.2, -0.8, 0.8, 2.2):
                which = ("bunny", "teapot", "treefir")[k % 3]
                ry = 0.7 * k
                if which == "bunny":
                    self.models.append(Model(name="bunny", position=[x, 0.32, z], rotation=[0, ry, 0], scale=0.8))
                elif which == "teapot":
                    self.models.append(Model(name="teapot", position=[x, 0.0, z], rotation=[0, ry, 0], scale=0.008))
                else:
                    self.models.append(Model(name="treefir", position=[x, 0.0, z], rotation=[0, ry, 0], scale=0.8))
                k += 1
        self.lights = [
            Light.spotLight(position=[2, 3, 4], direction=[-1.5, -2.5, -3.0], coneAngle=np.float32(35.0 / 180.0 * math.pi), color=[30, 30, 30]),
            Light.sunLight(direction=[-1, -2, -0.5], color=[1, 1, 1]),
        ]


SCENES = {"dragon": DragonScene, "dragon_irregular": IrregularDragonScene, "dragon_hostile": HostileDragonScene, "cornell": CornellScene, "dragon4": InstancedDragonScene, "garden": GardenScene}


def _geometry_key(mesh):
    return (mesh.positions.__array_interface__["data"][0], mesh.normals.__array_interface__["data"][0], mesh.positions.shape[0], tuple(id(s) for s in mesh.submeshes))


def flatten_scene(scene, share=False):
    """(positions, normals, transform16, [(indices, Material)]) per mesh, in instance order — the
    arrays that cross the C ABI (and that tests hand to the oracle as well).

    share=True: 5-tuples (..., source) where source is the index of an earlier mesh with the very same vertex arrays and
    submeshes (Model caches what it loads, so `Model(name="dragon")` x 4 is one geometry) or -1; such entries become
    mrt_scene_add_instance calls — the instances of BASELINE.json configs[4]."""
    out, first = [], {}
    for i, mesh in enumerate(scene.meshes):
        entry = (mesh.positions, mesh.normals, np.ascontiguousarray(mesh.transform.reshape(16)), [(s.indices, s.material) for s in mesh.submeshes])
        if share:
            src = first.setdefault(_geometry_key(mesh), i)
            entry = entry + (src if src != i else -1,)
        out.append(entry)
    return out
