"""world_size-2 (and 3) gloo tests of the multi-GPU path's host logic on CPU: the tile partition, the
sample partition and the one reduce that assembles the image.  The per-rank images come from the CPU
oracle here (there is no GPU); on the GPU box the same partition is checked against the HIP renderer
(test_gpu_parity.py::test_shards_sum_to_full_frame)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mode, w, h, frames, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import metal_raytracing_amd as m
    from metal_raytracing_amd import distributed as D
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = m.CornellScene((w, h))
    osc = O.OracleScene(m.flatten_scene(sc), sc.lights)
    r = O.OracleRenderer(osc, w, h, camera=sc.camera)
    if mode == "tile":
        r.set_shard(rank, world)
    else:
        r.set_sample_offset(rank * frames)
    r.render(frames, threads=2)
    acc = torch.from_numpy(r.accumulation().copy())
    own = D.tile_owner_map(w, h, world) == rank
    if mode == "tile":
        assert (acc.numpy()[~own] == 0).all() and int(own.sum()) == D.owned_pixel_count(w, h, rank, world)
    if mode == "tile":            # the compact assemble beside the reduce: every rank ships only the tiles it owns; same image bit for bit
        acc2 = acc.clone()
        D.gather_compact(acc2, dst=0)
    D.reduce_accumulation(acc, mode, dst=0)
    if rank == 0:
        np.save(out_path, acc.numpy())
        if mode == "tile": assert np.array_equal(acc.numpy().view(np.uint32), acc2.numpy().view(np.uint32)), "compact assemble != reduce"
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("tile", 2), ("tile", 3), ("sample", 2)])
def test_sharded_image_assembles(tmp_path, mode, world, mrt, orc):
    w, h, frames = 72, 40, 2
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, w, h, frames, out), nprocs=world, join=True)
    got = np.load(out)
    sc = mrt.CornellScene((w, h))
    osc = orc.OracleScene(mrt.flatten_scene(sc), sc.lights)
    if mode == "tile":
        ref = orc.OracleRenderer(osc, w, h, camera=sc.camera); ref.render(frames)
        assert np.array_equal(got, ref.accumulation())            # disjoint sum: exact
    else:
        means = []
        for rank in range(world):
            r = orc.OracleRenderer(osc, w, h, camera=sc.camera); r.set_sample_offset(rank * frames); r.render(frames)
            means.append(r.accumulation().astype(np.float64))
        assert np.allclose(got, sum(means) / world, rtol=1e-6, atol=1e-7)
        full = orc.OracleRenderer(osc, w, h, camera=sc.camera); full.render(frames * world)
        assert np.allclose(got[..., :3], full.accumulation()[..., :3], rtol=2e-5, atol=1e-6)   # same samples, different association


def test_tile_partition_properties(mrt):
    from metal_raytracing_amd import distributed as D
    for (w, h, world) in [(1920, 1080, 8), (37, 21, 3), (8, 8, 2), (3840, 2160, 8)]:
        owner = D.tile_owner_map(w, h, world)
        counts = [D.owned_pixel_count(w, h, r, world) for r in range(world)]
        assert sum(counts) == w * h and owner.min() == 0 and owner.max() == min(world, ((w + 7) // 8) * ((h + 7) // 8)) - 1
        if w * h > 100000:
            assert max(counts) - min(counts) <= 64 * 2                 # round-robin tiles balance to within a tile or two
