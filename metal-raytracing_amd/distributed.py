"""Multi-GPU sharding of one image: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI), no collective on the data path while rendering, ONE reduce of the RGBA32F radiance buffer per
output image.  The reference is single-GPU (MTLCreateSystemDefaultDevice,
Renderer.swift:46); this layer is new.

Two partitions:
  * "tile"   — rank r owns the 8x8 screen tiles with tile_id % world == r (round-robin, so the expensive
               screen regions are spread over all GPUs); every rank accumulates its own pixels over all
               frames, other pixels stay 0; the reduce is a disjoint sum.  Strong scaling.
  * "sample" — every rank renders full frames for a disjoint range of sample indices (Halton index offset);
               the reduce sums the per-rank means and rank 0 divides by world.  Weak scaling.
"""
import numpy as np
import torch
import torch.distributed as dist


DEFAULT_FRAME_BATCH = 8      # csrc/renderer.h DEFAULT_FRAME_BATCH: frames a pass carries at most on one GPU
MAX_FRAME_BATCH = 32         # csrc/renderer.h MAX_FRAME_BATCH


def auto_frame_batch(pixels, base=DEFAULT_FRAME_BATCH):
    """The library's default frames per pass (renderer option frame_batch = 0, csrc/renderer.hip Renderer::batch_wanted): a pass carries about as many pixel-frames as
    eight 1920 x 1080 frames — 8 at that size and above, proportionally more for a smaller image, at most 32."""
    ref = base * 1920 * 1080
    return min(MAX_FRAME_BATCH, max(base, -(-ref // max(1, int(pixels)))))


def shard_frame_batch(world, frames_total=None, base=DEFAULT_FRAME_BATCH):
    """Frames per pass of one rank of a `world`-GPU tile-sharded run.  A rank owns 1/world of the pixels of every frame, so it carries
    proportionally more frames per pass to keep its launches large (base x world, at most 32) — but never more than a third of the run's frames (and never
    fewer than the 1-GPU default): a short run keeps about three passes to overlap.  mrt_group_renderer_create applies the first half of the rule, and
    Renderer::render the second to every draw (csrc/renderer.hip `batch_cap`), so both launch paths agree."""
    if world <= 1:
        return base
    fb = min(MAX_FRAME_BATCH, base * world)
    if frames_total is not None:
        fb = min(fb, max(base, int(frames_total) // 3))
    return fb


def pass_sizes(n_frames, frame_batch, base=DEFAULT_FRAME_BATCH):
    """How Renderer::render (csrc/renderer.hip) splits a draw of n_frames into passes: passes larger than the default take at most a third of the draw,
    and the draw's frames go in ceil(n / batch) passes of (almost) equal size — 20 frames at frame_batch 8 = 7 + 7 + 6."""
    cap = min(frame_batch, max(base, (n_frames + 2) // 3)) if frame_batch > base else frame_batch
    n_passes = (n_frames + cap - 1) // cap
    out, f = [], 0
    for p in range(n_passes):
        b = min(cap, (n_frames - f + (n_passes - p) - 1) // max(1, n_passes - p))
        out.append(b); f += b
    return out


def tile_owner_map(width, height, world):
    """(height, width) int32: owning rank of every pixel under the "tile" partition."""
    tiles_x = (width + 7) // 8
    ys, xs = np.mgrid[0:height, 0:width]
    return (((ys // 8) * tiles_x + xs // 8) % world).astype(np.int32)


def owned_pixel_count(width, height, rank, world):
    return int((tile_owner_map(width, height, world) == rank).sum())


def reduce_accumulation(accum: torch.Tensor, mode="tile", dst=0, group=None):
    """Assemble the image on `dst` from the per-rank accumulation buffers (in place on `dst`)."""
    world = dist.get_world_size(group)
    dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if mode == "sample" and dist.get_rank(group) == dst:
        accum /= world
    return accum


def shard_tiles(width, height, rank, world):
    """8 x 8 tiles of the image that shard (rank, world) owns (tile_id % world == rank)."""
    tiles = ((width + 7) // 8) * ((height + 7) // 8)
    return max(0, (tiles - rank + world - 1) // world)


def _tile_index(width, height, rank, world):
    """(ys, xs, inside) of the compact buffer's tiles x 64 pixels: tile lt of the shard is tile lt * world + rank of the image, its 8 x 8 pixels row-major."""
    tl = shard_tiles(width, height, rank, world)
    tiles_x = (width + 7) // 8
    tile = np.arange(tl, dtype=np.int64)[:, None] * world + rank
    k = np.arange(64, dtype=np.int64)[None, :]
    ys, xs = (tile // tiles_x) * 8 + k // 8, (tile % tiles_x) * 8 + k % 8
    return ys, xs, (xs < width) & (ys < height)


def pack_owned_tiles(accum, rank, world):
    """Host twin of mrt_renderer_pack_owned_tiles (csrc/renderer.hip k_tiles): (h, w, 4) -> (tiles x 64, 4), pixels outside the image 0."""
    a = np.asarray(accum); h, w = a.shape[:2]
    ys, xs, inside = _tile_index(w, h, rank, world)
    out = np.zeros(ys.shape + (4,), a.dtype)
    out[inside] = a[ys[inside], xs[inside]]
    return out.reshape(-1, 4)


def unpack_tiles(image, compact, rank, world):
    """Host twin of mrt_renderer_unpack_tiles: writes shard (rank, world)'s compact buffer into the image (in place)."""
    h, w = image.shape[:2]
    ys, xs, inside = _tile_index(w, h, rank, world)
    c = np.asarray(compact).reshape(ys.shape + (4,))
    image[ys[inside], xs[inside]] = c[inside]
    return image


def gather_compact(accum: torch.Tensor, dst=0, group=None, pack=None, unpack=None):
    """The compact assemble of a tile-sharded image (beside reduce_accumulation): every rank ships only the tiles it owns — 1 / world of the image — as one
    dist.gather of equally sized (padded) compact buffers, and `dst` writes them in place.  Same image as the reduce, bit for bit (nothing is added).
    pack(rank) -> this rank's compact tensor / unpack(compact, rank): device kernels of a ShardedRenderer; default: the numpy twins on host tensors."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    h, w = accum.shape[:2]
    n_max = shard_tiles(w, h, 0, world) * 64                 # rank 0 owns the most tiles
    mine = pack(rank) if pack else torch.from_numpy(pack_owned_tiles(accum.numpy(), rank, world))
    send = torch.zeros((n_max, 4), dtype=accum.dtype, device=mine.device); send[: mine.shape[0]] = mine
    recv = [torch.zeros_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank == dst:
        for r in range(world):
            if r == dst: continue
            n = shard_tiles(w, h, r, world) * 64
            if unpack: unpack(recv[r][:n], r)
            else: unpack_tiles(accum.numpy(), recv[r][:n].numpy(), r, world)
    return accum


class ShardedRenderer:
    """A Renderer bound to this rank's GPU and shard; `gather()` runs the ONE collective per output image.  bench.py's N > 1
    path and the multi-process tests drive the multi-GPU case through this class (world == 1: a plain renderer, gather is a copy).

    backend None / "nccl": the reduce runs on the device buffer (RCCL over xGMI).  "gloo": rehearsal without a GPU per rank —
    ranks may share one device and the reduce runs on a host copy.
    """

    def __init__(self, size, scene, rank, world, mode="tile", device=None, frames_total=None, backend=None, **kw):
        from .renderer import Renderer
        if mode not in ("tile", "sample"):
            raise ValueError(mode)
        self.rank, self.world, self.mode, self.backend = rank, world, mode, backend
        self.renderer = Renderer(size, scene, device=rank if device is None else device, **kw)
        if world > 1:
            if mode == "tile":
                self.renderer.set_shard(rank, world)
                # 1/world of the pixels per frame: carry proportionally more frames per pass so that the launches stay large — but a short run
                # still needs about three passes to overlap (tools/tile_scaling_probe.py, rank 0 of N on one GPU, 20 frames: N = 8 at 32 frames
                # per pass 4.98, at 8 frames per pass 5.42 Grays/s per rank; 240 frames: 9.73 at 32, 8.66 at 8)
                self.renderer.set_option("frame_batch", shard_frame_batch(world, frames_total))
            else:
                if frames_total is None:
                    raise ValueError("sample sharding needs frames_total (frames per rank)")
                self.renderer.set_option("sample_offset", rank * int(frames_total))
        w, h = self.renderer.size
        self.buffer = torch.zeros((h, w, 4), dtype=torch.float32, device=f"cuda:{self.renderer.ctx.device}") if world > 1 else None

    def draw(self, frames=1):
        self.renderer.draw(frames)

    def gather(self, dst=0, compact=False):
        """The assembled image on `dst`; other ranks get their partial buffer.  Default: the ONE reduce(sum) of the whole RGBA32F buffers (north_star).
        compact=True (tile mode): every rank ships only the tiles it owns, packed and unpacked by the renderer's own kernels (gather_compact)."""
        r = self.renderer
        if self.world == 1:
            return torch.from_numpy(r.accumulation())
        if compact and self.mode == "tile":
            return self._gather_compact(dst)
        r.copy_accum_to(self.buffer.data_ptr(), self.buffer.numel() * 4)
        r.wait()
        if self.backend == "gloo":
            host = self.buffer.cpu()
            reduce_accumulation(host, self.mode, dst)
            self.buffer.copy_(host)
        else:
            reduce_accumulation(self.buffer, self.mode, dst)
        return self.buffer

    def _gather_compact(self, dst):
        r = self.renderer; w, h = r.size
        dev = self.buffer.device
        def pack(rank):
            t = torch.empty((r.shard_tiles(rank, self.world) * 64, 4), dtype=torch.float32, device=dev)
            r.pack_owned_tiles(t.data_ptr(), t.numel() * 4); r.wait()
            return t.cpu() if self.backend == "gloo" else t
        def unpack(compact, rank):
            c = compact.to(dev).contiguous()
            r.unpack_tiles(c.data_ptr(), c.numel() * 4, rank, self.world); r.wait()
        r.wait()
        shape = torch.empty((h, w, 4), dtype=torch.float32, device="cpu" if self.backend == "gloo" else dev)          # (only its shape is read when pack / unpack are given)
        gather_compact(shape, dst, pack=pack, unpack=unpack)
        r.copy_accum_to(self.buffer.data_ptr(), self.buffer.numel() * 4); r.wait()          # the root's accumulation buffer now holds the whole image
        return self.buffer

    def close(self):
        self.renderer.close()
