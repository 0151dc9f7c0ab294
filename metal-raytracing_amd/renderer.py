"""Host-side mirror of Renderer (Renderer.swift:12-357) over the C ABI.

`Renderer(size, scene)` does what `Renderer.init?(metalView:)` does (device + queue, scene,
targets, buffers, acceleration structures); `draw()` is `draw(in:)` — one frame: uniforms,
dispatch, ping-pong swap.  There is no MTKView: the tonemap pass writes RGBA8 to host memory
instead of a drawable (`tonemapped()`).
"""
import time
import ctypes as C

import numpy as np

from ._ffi import Camera, Light, MRTError, RenderStats, SceneStats, check, lib, ptr
from .scene import DragonScene, Scene


class Context:
    """MTLCreateSystemDefaultDevice + makeCommandQueue (Renderer.swift:46-59)."""

    def __init__(self, device=0):
        self.handle = C.c_void_p()
        check(lib.mrt_context_create(int(device), C.byref(self.handle)))
        self.device = int(device)

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        check(lib.mrt_context_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def set_stream(self, hip_stream):
        check(lib.mrt_context_set_stream(self.handle, C.c_void_p(hip_stream) if hip_stream else None))

    def close(self):
        if self.handle:
            lib.mrt_context_destroy(self.handle)
            self.handle = C.c_void_p()


class DeviceScene:
    """The committed device-side scene: geometry upload + createAccelerationStructures
    (Renderer.swift:184-214)."""

    def __init__(self, ctx, scene, options=None):
        self.ctx = ctx
        self.handle = C.c_void_p()
        check(lib.mrt_scene_create(ctx.handle, C.byref(self.handle)))
        for k, v in (options or {}).items():
            check(lib.mrt_scene_set_option(self.handle, k.encode(), float(v)))
        from .scene import flatten_scene
        for pos, nrm, xf, subs, source in flatten_scene(scene, share=True):
            mid = C.c_int32()
            if source >= 0:                                   # same geometry as an earlier mesh: an instance of it
                check(lib.mrt_scene_add_instance(self.handle, source, ptr(xf), C.byref(mid)))
                continue
            pos = np.ascontiguousarray(pos, np.float32)
            nrm = np.ascontiguousarray(nrm, np.float32)
            check(lib.mrt_scene_add_mesh(self.handle, ptr(pos), 12, ptr(nrm), 12, pos.shape[0], ptr(xf), C.byref(mid)))
            for idx, mat in subs:
                idx = np.ascontiguousarray(idx, np.uint32)
                check(lib.mrt_mesh_add_submesh(self.handle, mid.value, ptr(idx), idx.shape[0], C.byref(mat), None))
        self.set_lights(scene.lights)
        t0 = time.perf_counter()
        check(lib.mrt_scene_commit(self.handle))
        self.commit_wall_ms = (time.perf_counter() - t0) * 1e3      # host wall time of the commit (uploads, build, validation); stats.build_ms is the device time of the build alone

    def set_instance_transform(self, mesh_id, transform):
        """Animated transforms: new object->world matrix for one instance; call commit() afterwards."""
        xf = np.ascontiguousarray(np.asarray(transform, np.float32).reshape(16))
        check(lib.mrt_scene_set_instance_transform(self.handle, int(mesh_id), ptr(xf)))

    def update_mesh(self, mesh_id, positions, normals):
        """Deforming geometry: new object-space positions / normals of one mesh's vertices (same count); call commit() afterwards — a flattened scene refits its tree."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3); nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        if nrm.shape[0] != pos.shape[0]:          # the C side reads vertex_count normals: a shorter array would be read past its end
            raise MRTError(1, f"update_mesh: {nrm.shape[0]} normals for {pos.shape[0]} positions (one normal per vertex)")
        check(lib.mrt_scene_update_mesh(self.handle, int(mesh_id), ptr(pos), 12, ptr(nrm), 12, pos.shape[0]))

    @property
    def refits(self):
        v = C.c_uint32()
        check(lib.mrt_debug_scene_refits(self.handle, C.byref(v)))
        return v.value

    def commit(self):
        check(lib.mrt_scene_commit(self.handle))

    def set_lights(self, lights):
        arr = (Light * max(1, len(lights)))(*lights)
        check(lib.mrt_scene_set_lights(self.handle, arr, len(lights)))

    @property
    def stats(self):
        s = SceneStats()
        check(lib.mrt_scene_stats(self.handle, C.byref(s)))
        return s

    def intersect_closest(self, rays):
        """rays: (n, 8) float32 [ox,oy,oz,tmin,dx,dy,dz,tmax] → structured array of Intersection."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=INTERSECTION_DTYPE)
        check(lib.mrt_scene_intersect_closest(self.handle, ptr(rays), rays.shape[0], ptr(out)))
        return out

    def intersect_any(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=np.int32)
        check(lib.mrt_scene_intersect_any(self.handle, ptr(rays), rays.shape[0], ptr(out)))
        return out

    def intersect_stream(self, rays, any_hit=False):
        """The render kernels' traversal (8-wide stream, both levels of an instanced scene) on caller rays; min_distance must be 0."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=INTERSECTION_DTYPE)
        check(lib.mrt_debug_intersect_stream(self.handle, ptr(rays), rays.shape[0], 1 if any_hit else 0, ptr(out)))
        return out

    def traversal_stats(self, rays, any_hit=False, alu_dup=0, mem_dup=0):
        """Diagnostics: (n, 8) uint32 {node visits, leaf visits, triangle tests, hit gid, t0, t1, 0, 0} per ray."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros((rays.shape[0], 8), np.uint32)
        check(lib.mrt_debug_traversal_stats(self.handle, ptr(rays), rays.shape[0], (1 if any_hit else 0) | (alu_dup << 8) | (mem_dup << 16), ptr(out)))
        return out

    def stream_stats(self, rays, any_hit=False, per_wave=256):
        """Diagnostics: (nwaves, 8) uint32 lane accounting of the wide stream traversal."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        nw = (rays.shape[0] + per_wave - 1) // per_wave
        out = np.zeros((nw, 8), np.uint32)
        check(lib.mrt_debug_stream_stats(self.handle, ptr(rays), rays.shape[0], 1 if any_hit else 0, per_wave, ptr(out), nw))
        return out

    def close(self):
        if self.handle:
            lib.mrt_scene_destroy(self.handle)
            self.handle = C.c_void_p()


INTERSECTION_DTYPE = np.dtype([("type", np.int32), ("distance", np.float32), ("instance_id", np.int32), ("geometry_id", np.int32),
                               ("primitive_id", np.int32), ("u", np.float32), ("v", np.float32), ("_pad", np.int32)])


class Renderer:
    """Renderer.swift:12-357."""

    @property
    def maxFramesInFlight(self):
        """Renderer.maxFramesInFlight (Renderer.swift:33 keeps 3).  Here: passes in flight on separate HIP streams, each carrying
        `frame_batch` frames (library defaults: 6 passes in flight x 8 frames at 1080p and above, up to 32 frames per pass for smaller images); set with set_option("frames_in_flight", n)."""
        return int(self.get_option("frames_in_flight"))

    def __init__(self, size, scene=None, device=0, seed=1, max_bounces=3, ctx=None, scene_options=None):
        self.size = (int(size[0]), int(size[1]))
        self.scene = scene if scene is not None else DragonScene(self.size)   # Renderer.swift:61
        self._own_ctx = ctx is None
        self.ctx = ctx or Context(device)
        self.device_scene = DeviceScene(self.ctx, self.scene, scene_options)
        self.handle = C.c_void_p()
        check(lib.mrt_renderer_create(self.ctx.handle, self.device_scene.handle, self.size[0], self.size[1], int(seed), int(max_bounces), C.byref(self.handle)))
        self.max_bounces = int(max_bounces)
        self.set_camera(self.scene.camera)

    # -- Renderer.frameIndex (Renderer.swift:41)
    @property
    def frameIndex(self):
        v = C.c_uint32()
        check(lib.mrt_renderer_frame_index(self.handle, C.byref(v)))
        return v.value

    @frameIndex.setter
    def frameIndex(self, v):
        check(lib.mrt_renderer_set_frame_index(self.handle, int(v)))

    def set_camera(self, camera: Camera):
        check(lib.mrt_renderer_set_camera(self.handle, C.byref(camera)))

    # -- Renderer.uniforms / updateUniforms (Renderer.swift:216-229): size, frameIndex, lightCount and camera in one 96-byte block
    @property
    def uniforms(self):
        from ._ffi import Uniforms
        u = Uniforms()
        check(lib.mrt_renderer_get_uniforms(self.handle, C.byref(u)))
        return u

    @uniforms.setter
    def uniforms(self, u):
        check(lib.mrt_renderer_set_uniforms(self.handle, C.byref(u)))
        self.size = (int(u.width), int(u.height))

    @property
    def framesCompleted(self):
        """Frames whose accumulation has finished on the device (never blocks) — the poll form of the completion handler of
        Renderer.swift:285-287."""
        v = C.c_uint64()
        check(lib.mrt_renderer_frames_completed(self.handle, C.byref(v)))
        return v.value

    # the host's knobs (include/mrt_abi.h mrt_renderer_set_option); every other key is one of the library's A/B switches (mrt_debug_renderer_set_option: tests, tools/, bench.py --opt)
    PUBLIC_OPTIONS = ("max_bounces", "frames_in_flight", "sample_offset", "frame_batch", "megakernel", "materials", "lanes_used", "lane_bytes")

    def set_option(self, key, value):
        fn = lib.mrt_renderer_set_option if key in self.PUBLIC_OPTIONS else lib.mrt_debug_renderer_set_option
        check(fn(self.handle, key.encode(), float(value)))

    def get_option(self, key):
        v = C.c_double(0.0)
        fn = lib.mrt_renderer_get_option if key in self.PUBLIC_OPTIONS else lib.mrt_debug_renderer_get_option
        check(fn(self.handle, key.encode(), C.byref(v)))
        return v.value

    def set_shard(self, rank, world):
        check(lib.mrt_renderer_set_shard(self.handle, int(rank), int(world)))

    def drawableSizeWillChange(self, size):                   # Renderer.swift:353-356
        self.size = (int(size[0]), int(size[1]))
        check(lib.mrt_renderer_resize(self.handle, self.size[0], self.size[1]))
        self.scene.updateUniforms(self.size)
        self.set_camera(self.scene.camera)

    def draw(self, frames=1, wait=False):                     # Renderer.swift:284-351
        check(lib.mrt_renderer_render(self.handle, int(frames)))
        if wait:
            self.wait()

    def wait(self):
        check(lib.mrt_renderer_wait(self.handle))

    def accumulation(self):
        """accumulationTargets[0] (Renderer.swift:332-334): (h, w, 4) float32, row 0 = bottom of the image."""
        out = np.empty((self.size[1], self.size[0], 4), np.float32)
        check(lib.mrt_renderer_read_accum(self.handle, ptr(out), out.nbytes))
        return out

    def tonemapped(self):
        """fragmentShader (Shaders.metal:39-52): (h, w, 4) uint8, top row first."""
        out = np.empty((self.size[1], self.size[0], 4), np.uint8)
        check(lib.mrt_renderer_read_tonemapped_rgba8(self.handle, ptr(out), out.nbytes))
        return out

    def copy_accum_to(self, device_ptr, nbytes):
        check(lib.mrt_renderer_copy_accum_to_device(self.handle, C.c_void_p(device_ptr), nbytes))

    def write_accum_from(self, device_ptr, nbytes):
        check(lib.mrt_renderer_write_accum_from_device(self.handle, C.c_void_p(device_ptr), nbytes))

    def shard_tiles(self, rank, world):
        """8 x 8 tiles of this image that shard (rank, world) owns (a compact buffer of that shard is tiles x 64 RGBA32F pixels)."""
        n = C.c_uint64()
        check(lib.mrt_renderer_shard_tiles(self.handle, int(rank), int(world), C.byref(n)))
        return n.value

    def pack_owned_tiles(self, device_ptr, nbytes):
        check(lib.mrt_renderer_pack_owned_tiles(self.handle, C.c_void_p(device_ptr), nbytes))

    def unpack_tiles(self, device_ptr, nbytes, rank, world):
        check(lib.mrt_renderer_unpack_tiles(self.handle, C.c_void_p(device_ptr), nbytes, int(rank), int(world)))

    @property
    def stats(self):
        s = RenderStats()
        check(lib.mrt_renderer_stats(self.handle, C.byref(s)))
        return s

    def reset_stats(self):
        check(lib.mrt_renderer_reset_stats(self.handle))

    @property
    def kernel_times(self):
        """{class: (summed ms, launches)} over the launches of the last draw that carried their own start/stop events."""
        from ._ffi import KERNEL_CLASSES, KernelTimes
        k = KernelTimes()
        check(lib.mrt_renderer_kernel_times(self.handle, C.byref(k)))
        return {name: (float(k.ms[i]), int(k.launches[i])) for i, name in enumerate(KERNEL_CLASSES)}

    def close(self):
        if self.handle:
            lib.mrt_renderer_destroy(self.handle)
            self.handle = C.c_void_p()
        self.device_scene.close()
        if self._own_ctx:
            self.ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class _TemplateScene(DeviceScene):
    """A scene that is filled (meshes, lights, options) but not committed: the template a device group replicates."""

    def __init__(self, ctx_handle, scene, options=None):
        self.ctx = None
        self.handle = C.c_void_p()
        check(lib.mrt_scene_create(ctx_handle, C.byref(self.handle)))
        for k, v in (options or {}).items():
            check(lib.mrt_scene_set_option(self.handle, k.encode(), float(v)))
        from .scene import flatten_scene
        for pos, nrm, xf, subs, source in flatten_scene(scene, share=True):
            mid = C.c_int32()
            if source >= 0:
                check(lib.mrt_scene_add_instance(self.handle, source, ptr(xf), C.byref(mid)))
                continue
            pos = np.ascontiguousarray(pos, np.float32)
            nrm = np.ascontiguousarray(nrm, np.float32)
            check(lib.mrt_scene_add_mesh(self.handle, ptr(pos), 12, ptr(nrm), 12, pos.shape[0], ptr(xf), C.byref(mid)))
            for idx, mat in subs:
                idx = np.ascontiguousarray(idx, np.uint32)
                check(lib.mrt_mesh_add_submesh(self.handle, mid.value, ptr(idx), idx.shape[0], C.byref(mat), None))
        self.set_lights(scene.lights)


class GroupRenderer:
    """Renderer over the n GPUs of one node in ONE process (mrt_group_*): the scene is replicated, the image sharded by 8x8 screen tile
    (tile_id % n == rank), `gather()` runs the one reduce(sum) per output image (RCCL over xGMI; peer copies + add when a device is
    named twice).  The reference creates a single MTLDevice (Renderer.swift:46-59); this widens that seam."""

    def __init__(self, size, scene, devices, seed=1, max_bounces=3, scene_options=None):
        self.size = (int(size[0]), int(size[1]))
        self.scene = scene
        ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        self.group = C.c_void_p()
        check(lib.mrt_group_create(ids, len(devices), C.byref(self.group)))
        self.handle = C.c_void_p()
        self._template = None
        try:
            c0 = C.c_void_p()
            check(lib.mrt_group_context(self.group, 0, C.byref(c0)))
            self._template = _TemplateScene(c0, scene, scene_options)
            check(lib.mrt_group_renderer_create(self.group, self._template.handle, self.size[0], self.size[1], int(seed), int(max_bounces), C.byref(self.handle)))
            check(lib.mrt_group_set_camera(self.handle, C.byref(scene.camera)))
        except Exception:
            self.close()
            raise

    @property
    def world(self):
        n = C.c_int32()
        check(lib.mrt_group_size(self.group, C.byref(n)))
        return n.value

    @property
    def reduce_mode(self):
        m = C.c_int32(); buf = C.create_string_buffer(256)
        check(lib.mrt_group_reduce_mode(self.group, C.byref(m), buf, 256))
        return m.value, buf.value.decode()

    def set_reduce_mode(self, mode):
        check(lib.mrt_group_set_reduce_mode(self.group, int(mode)))

    def set_option(self, key, value):
        if key in Renderer.PUBLIC_OPTIONS:
            check(lib.mrt_group_set_option(self.handle, key.encode(), float(value)))
            return
        for rank in range(self.world):                       # an A/B switch: through every device's renderer
            r = C.c_void_p()
            check(lib.mrt_group_renderer_rank(self.handle, rank, C.byref(r)))
            check(lib.mrt_debug_renderer_set_option(r, key.encode(), float(value)))

    def rank_option(self, rank, key):
        r = C.c_void_p(); v = C.c_double()
        check(lib.mrt_group_renderer_rank(self.handle, int(rank), C.byref(r)))
        check(lib.mrt_debug_renderer_get_option(r, key.encode(), C.byref(v)))
        return v.value

    def draw(self, frames=1, wait=False):
        check(lib.mrt_group_render(self.handle, int(frames)))
        if wait:
            self.wait()

    def wait(self):
        check(lib.mrt_group_wait(self.handle))

    def rank_stats(self, rank):
        """One device's own counters and the device time of its last draw (mrt_renderer_stats of that rank's renderer): what a scaling record is diagnosed with."""
        r = C.c_void_p(); s = RenderStats()
        check(lib.mrt_group_renderer_rank(self.handle, int(rank), C.byref(r)))
        check(lib.mrt_renderer_stats(r, C.byref(s)))
        return s

    @property
    def framesCompleted(self):
        v = C.c_uint64()
        check(lib.mrt_group_frames_completed(self.handle, C.byref(v)))
        return v.value

    def gather(self, to_host=True):
        """The assembled image: (h, w, 4) float32, row 0 = bottom (None with to_host=False: it stays on the root device)."""
        if not to_host:
            check(lib.mrt_group_gather(self.handle, None, 0))
            return None
        out = np.empty((self.size[1], self.size[0], 4), np.float32)
        check(lib.mrt_group_gather(self.handle, ptr(out), out.nbytes))
        return out

    @property
    def stats(self):
        s = RenderStats()
        check(lib.mrt_group_stats(self.handle, C.byref(s)))
        return s

    def close(self):
        if self.handle:
            lib.mrt_group_renderer_destroy(self.handle)
            self.handle = C.c_void_p()
        if self._template is not None:
            self._template.close()
            self._template = None
        if self.group:
            lib.mrt_group_destroy(self.group)
            self.group = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def save_png(path, rgba8):
    """The tonemapped image (Renderer.tonemapped(): (h, w, 4) uint8, row 0 = top — the blit's flip, Shaders.metal:35) as an 8-bit RGBA PNG; zlib only."""
    import struct, zlib
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4: raise ValueError("save_png: expected (h, w, 4) uint8")
    h, w = a.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 4)], axis=1).tobytes()          # filter byte 0 in front of every row
    def chunk(tag, data): return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def save_pfm(path, accum):
    """The accumulation buffer (Renderer.accumulation(): (h, w, 4) float32 radiance, row 0 = the BOTTOM of the image, SURVEY a-4) as a little-endian colour PFM —
    whose rows are stored bottom to top as well, so the buffer goes out as it lies.  For comparing radiance, not display (no tonemap)."""
    a = np.asarray(accum, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] not in (3, 4): raise ValueError("save_pfm: expected (h, w, 3 or 4) float32")
    h, w = a.shape[:2]
    with open(path, "wb") as f:
        f.write(f"PF\n{w} {h}\n-1.0\n".encode()); f.write(np.ascontiguousarray(a[:, :, :3]).astype("<f4").tobytes())
