"""ctypes wrapper of the CPU oracle (oracle/_build/libmrt_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(_REPO, "oracle")
# MRT_ORACLE_NATIVE=1 (set by bench.py's cpu_baseline leg before the import): the -O3 -march=native build, made on the box it runs on
ORACLE_NATIVE = os.environ.get("MRT_ORACLE_NATIVE", "") == "1"
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libmrt_oracle_native.so" if ORACLE_NATIVE else "libmrt_oracle.so")
ORACLE_BUILD = "-O3 -march=native" if ORACLE_NATIVE else "-O2 -march=x86-64-v3"


def build_oracle(force=False):
    src = os.path.join(ORACLE_DIR, "mrt_oracle.cpp")
    if force or not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"] + (["-B"] if force else []) + (["native"] if ORACLE_NATIVE else []))      # force: really recompile (-march=native is only valid on the box that compiled it)
    return ORACLE_LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        l = C.CDLL(ORACLE_LIB)
        P, F, I, U = C.c_void_p, C.c_float, C.c_int, C.c_uint32
        l.orc_scene_create.restype = P
        l.orc_scene_destroy.argtypes = [P]
        l.orc_scene_add_mesh.argtypes = [P, P, P, C.c_size_t, P]
        l.orc_mesh_add_submesh.argtypes = [P, I, P, C.c_size_t, P]
        l.orc_scene_set_lights.argtypes = [P, P, I]
        l.orc_scene_commit.argtypes = [P]
        l.orc_scene_add_instance.argtypes = [P, I, P]
        l.orc_scene_set_instancing.argtypes = [P, I]
        l.orc_scene_set_transform.argtypes = [P, I, P]
        l.orc_scene_triangles.argtypes = [P]; l.orc_scene_triangles.restype = C.c_uint64
        l.orc_scene_nodes.argtypes = [P]; l.orc_scene_nodes.restype = C.c_uint64
        l.orc_intersect_closest.argtypes = [P, P, C.c_size_t, P, I]
        l.orc_intersect_any.argtypes = [P, P, C.c_size_t, P, I]
        l.orc_renderer_create.restype = P; l.orc_renderer_create.argtypes = [P, I, I, U, I]
        l.orc_renderer_destroy.argtypes = [P]
        l.orc_renderer_set_camera.argtypes = [P, P]
        l.orc_renderer_set_shard.argtypes = [P, I, I]
        l.orc_renderer_set_frame_index.argtypes = [P, U]
        l.orc_renderer_set_sample_offset.argtypes = [P, U]
        l.orc_renderer_set_materials.argtypes = [P, I]
        l.orc_renderer_set_accum.argtypes = [P, P]
        l.orc_renderer_render.argtypes = [P, I, I, I, P]
        l.orc_renderer_read_accum.argtypes = [P, P]
        l.orc_renderer_counters.argtypes = [P, P, P]
        l.orc_tonemap_rgba8.argtypes = [P, I, I, P]
        l.orc_halton.restype = F; l.orc_halton.argtypes = [I, I]
        l.orc_sincos_2pi.argtypes = [F, P, P]
        l.orc_hemisphere.argtypes = [F, F, P]
        l.orc_align.argtypes = [P, P, P]
        l.orc_seed_hash.restype = U; l.orc_seed_hash.argtypes = [U, U]
        l.orc_make_transform.argtypes = [P, P, F, P]
        l.orc_default_camera.argtypes = [I, I, P]
        l.orc_sample_area_light.argtypes = [P, P, P, P, P, P]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """Built from the same arrays that cross the product's C ABI (metal_raytracing_amd.flatten_scene)."""

    def __init__(self, meshes, lights, instancing=False):
        """meshes: flatten_scene(scene) 4-tuples, or flatten_scene(scene, share=True) 5-tuples whose last field names the mesh an
        entry is an instance of.  instancing=True: the two-level restatement (object-space intersection per instance)."""
        l = lib()
        self.h = C.c_void_p(l.orc_scene_create())
        self._keep = []
        l.orc_scene_set_instancing(self.h, 1 if instancing else 0)
        for entry in meshes:
            pos, nrm, xf, subs = entry[:4]
            xf = np.ascontiguousarray(xf, np.float32)
            if len(entry) > 4 and entry[4] >= 0:
                l.orc_scene_add_instance(self.h, int(entry[4]), _p(xf))
                continue
            pos = np.ascontiguousarray(pos, np.float32); nrm = np.ascontiguousarray(nrm, np.float32)
            mid = l.orc_scene_add_mesh(self.h, _p(pos), _p(nrm), pos.shape[0], _p(xf))
            for idx, mat in subs:
                idx = np.ascontiguousarray(idx, np.uint32)
                l.orc_mesh_add_submesh(self.h, mid, _p(idx), idx.shape[0], C.byref(mat))
        n = len(lights)
        if n:
            arr = (type(lights[0]) * n)(*lights)
            l.orc_scene_set_lights(self.h, C.byref(arr), n)
        l.orc_scene_commit(self.h)

    def set_transform(self, mesh_id, xf16):
        xf = np.ascontiguousarray(np.asarray(xf16, np.float32).reshape(16))
        lib().orc_scene_set_transform(self.h, int(mesh_id), _p(xf))
        lib().orc_scene_commit(self.h)

    @property
    def triangles(self):
        return int(lib().orc_scene_triangles(self.h))

    def intersect_closest(self, rays, brute=False):
        from metal_raytracing_amd import INTERSECTION_DTYPE
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=INTERSECTION_DTYPE)
        lib().orc_intersect_closest(self.h, _p(rays), rays.shape[0], _p(out), 1 if brute else 0)
        return out

    def intersect_any(self, rays, brute=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=np.int32)
        lib().orc_intersect_any(self.h, _p(rays), rays.shape[0], _p(out), 1 if brute else 0)
        return out

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h); self.h = None


class OracleRenderer:
    def __init__(self, scene: OracleScene, width, height, seed=1, max_bounces=3, camera=None):
        self.scene, self.w, self.h_, self.max_bounces = scene, width, height, max_bounces
        self.h = C.c_void_p(lib().orc_renderer_create(scene.h, width, height, seed, max_bounces))
        if camera is not None:
            lib().orc_renderer_set_camera(self.h, C.byref(camera))

    def set_shard(self, rank, world):
        lib().orc_renderer_set_shard(self.h, rank, world)

    def set_frame_index(self, fi):
        lib().orc_renderer_set_frame_index(self.h, fi)

    def set_sample_offset(self, so):
        lib().orc_renderer_set_sample_offset(self.h, so)

    def set_materials(self, on=True):
        lib().orc_renderer_set_materials(self.h, 1 if on else 0)

    def render(self, frames=1, threads=0, brute=False, dump=False):
        d = None
        if dump:
            d = np.zeros((self.h_, self.w, self.max_bounces, 16), np.float32)
        lib().orc_renderer_render(self.h, frames, threads, 1 if brute else 0, _p(d) if dump else None)
        return d

    def accumulation(self):
        out = np.empty((self.h_, self.w, 4), np.float32)
        lib().orc_renderer_read_accum(self.h, _p(out))
        return out

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        lib().orc_renderer_counters(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if self.h:
            lib().orc_renderer_destroy(self.h); self.h = None


def tonemap_rgba8(rgba):
    h, w = rgba.shape[:2]
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.empty((h, w, 4), np.uint8)
    lib().orc_tonemap_rgba8(_p(rgba), w, h, _p(out))
    return out


def halton(i, d):
    return float(lib().orc_halton(int(i), int(d)))


def sincos_2pi(u):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos_2pi(float(u), C.byref(s), C.byref(c))
    return s.value, c.value


def hemisphere(ux, uy):
    out = np.zeros(3, np.float32)
    lib().orc_hemisphere(float(ux), float(uy), _p(out))
    return out


def align(s, n):
    s = np.ascontiguousarray(s, np.float32); n = np.ascontiguousarray(n, np.float32)
    out = np.zeros(3, np.float32)
    lib().orc_align(_p(s), _p(n), _p(out))
    return out


def seed_hash(seed, idx):
    return int(lib().orc_seed_hash(seed, idx))


def make_transform(p, r, s):
    p = np.ascontiguousarray(p, np.float32); r = np.ascontiguousarray(r, np.float32)
    out = np.zeros(16, np.float32)
    lib().orc_make_transform(_p(p), _p(r), float(s), _p(out))
    return out.reshape(4, 4)


def default_camera(w, h):
    from metal_raytracing_amd import Camera
    c = Camera()
    lib().orc_default_camera(w, h, C.byref(c))
    return c


def sample_area_light(light, u2, pos):
    u2 = np.ascontiguousarray(u2, np.float32); pos = np.ascontiguousarray(pos, np.float32)
    d = np.zeros(3, np.float32); col = np.zeros(3, np.float32); dist = C.c_float()
    lib().orc_sample_area_light(C.byref(light), _p(u2), _p(pos), _p(d), _p(col), C.byref(dist))
    return d, col, dist.value
