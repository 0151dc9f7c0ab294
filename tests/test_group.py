"""Device group (mrt_group_*, include/mrt_abi.h): one process, n devices, replicated scene, tile_id % n shards, ONE reduce(sum) per output
image.  The GPU box has one GPU, so n > 1 groups name device 0 several times: everything but the ncclReduce call itself runs (RCCL refuses
duplicate devices; such a group assembles with peer copies + add, which is also selectable on a real multi-GPU group)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _plain(mrt, ctx, sc, w, h, frames, bounces=3):
    r = mrt.Renderer((w, h), sc, ctx=ctx, max_bounces=bounces)
    r.draw(frames, wait=True)
    a, st = r.accumulation(), r.stats
    r.close()
    return a, st


@pytest.mark.parametrize("n", [1, 2, 3])
def test_group_image_is_bit_identical_to_one_device(mrt, orc, gpu_ctx, n):
    from test_gpu_parity import assert_parity, oracle_render
    w, h, frames = 200, 120, 6
    sc = mrt.CornellScene((w, h))
    ref, st = _plain(mrt, gpu_ctx, sc, w, h, frames)
    with mrt.GroupRenderer((w, h), sc, [0] * n) as g:
        assert g.world == n
        mode, note = g.reduce_mode
        assert mode == 1 and (("one device" in note) if n == 1 else ("more than once" in note))
        assert g.rank_option(0, "frame_batch") == (8 if n == 1 else min(32, 8 * n))
        g.draw(frames)
        img = g.gather()
        assert g.framesCompleted == frames
        gs = g.stats
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert (gs.closest_rays, gs.shadow_rays, gs.primary_rays, gs.frames) == (st.closest_rays, st.shadow_rays, st.primary_rays, frames)
    oimg, _ = oracle_render(orc, mrt, sc, w, h, frames)
    assert_parity(img, oimg, exact_frac=1.0)


@pytest.mark.parametrize("n", [2, 3])
def test_group_compact_assemble_is_the_reduce(mrt, gpu_ctx, n):
    """Reduce mode 2: every rank packs the tiles it owns (1 / n of the image) and the root writes them in place — peer copies here (the group names one device n times), ncclSend /
    ncclRecv on a real multi-GPU group.  Same image as peer copies + add and as one device, bit for bit; the device pack is the host twin's (distributed.pack_owned_tiles)."""
    import torch
    from metal_raytracing_amd import distributed as D
    w, h, frames = 203, 117, 5          # ragged: partial edge tiles
    sc = mrt.CornellScene((w, h))
    ref, st = _plain(mrt, gpu_ctx, sc, w, h, frames)
    with mrt.GroupRenderer((w, h), sc, [0] * n) as g:
        g.draw(frames)
        summed = g.gather()
        g.set_reduce_mode(2)
        mode, note = g.reduce_mode
        assert mode == 2 and "compact" in note and "peer copies" in note
        packed = g.gather(); again = g.gather()
    assert np.array_equal(summed.view(np.uint32), ref.view(np.uint32)) and np.array_equal(packed.view(np.uint32), ref.view(np.uint32)) and np.array_equal(again, packed)
    # one shard's renderer: pack on the device == the host twin of its accumulation buffer; unpack into a second renderer puts the tiles back
    r = mrt.Renderer((w, h), sc, ctx=gpu_ctx); r.set_shard(1, n); r.draw(frames, wait=True)
    acc = r.accumulation()
    tl = r.shard_tiles(1, n); assert tl == D.shard_tiles(w, h, 1, n)
    t = torch.empty((tl * 64, 4), dtype=torch.float32, device="cuda:0")
    r.pack_owned_tiles(t.data_ptr(), t.numel() * 4); r.wait()
    assert np.array_equal(t.cpu().numpy(), D.pack_owned_tiles(acc, 1, n))
    r2 = mrt.Renderer((w, h), sc, ctx=gpu_ctx); r2.draw(1, wait=True); base = r2.accumulation().copy()
    r2.unpack_tiles(t.data_ptr(), t.numel() * 4, 1, n); r2.wait()
    own = D.tile_owner_map(w, h, n) == 1
    got = r2.accumulation()
    assert np.array_equal(got[own], acc[own]) and np.array_equal(got[~own], base[~own])
    with pytest.raises(mrt.MRTError): r.pack_owned_tiles(t.data_ptr(), t.numel() * 4 - 16)
    r.close(); r2.close()


def test_group_accumulates_across_calls_and_options_reach_every_rank(mrt, gpu_ctx):
    w, h = 96, 64
    sc = mrt.SCENES["dragon_small"]((w, h)) if "dragon_small" in mrt.SCENES else mrt.CornellScene((w, h))
    ref, _ = _plain(mrt, gpu_ctx, sc, w, h, 5, bounces=4)
    with mrt.GroupRenderer((w, h), sc, [0, 0], max_bounces=4) as g:
        g.set_option("frames_in_flight", 3)
        assert g.rank_option(1, "frames_in_flight") == 3
        g.draw(2); g.draw(3)
        a = g.gather()
        b = g.gather()                       # gathering twice does not change the image
        g.gather(to_host=False)
    assert np.array_equal(a, ref) and np.array_equal(a, b)


def test_group_rejects_bad_arguments(mrt):
    import ctypes as C
    from metal_raytracing_amd._ffi import lib
    g = C.c_void_p()
    assert lib.mrt_group_create(None, 1, C.byref(g)) != 0
    ids = (C.c_int32 * 1)(99)
    assert lib.mrt_group_create(ids, 1, C.byref(g)) != 0 and not g
    ids = (C.c_int32 * 1)(0)
    assert lib.mrt_group_create(ids, 0, C.byref(g)) != 0
    assert lib.mrt_group_create(ids, 1, C.byref(g)) == 0
    assert lib.mrt_group_set_reduce_mode(g, 7) != 0
    assert lib.mrt_group_destroy(g) == 0


def test_one_device_group_through_rccl(mrt, gpu_ctx):
    """The RCCL path itself — dlopen of librccl, ncclCommInitAll, ncclGroupStart / ncclReduce(sum, float32, root 0) / ncclGroupEnd on the renderer's stream — on the one GPU
    this box has: a reduce over one rank is a copy, so the gathered image must equal the renderer's."""
    w, h = 160, 96
    sc = mrt.CornellScene((w, h))
    ref, _ = _plain(mrt, gpu_ctx, sc, w, h, 3)
    with mrt.GroupRenderer((w, h), sc, [0]) as g:
        g.set_reduce_mode(0)
        mode, note = g.reduce_mode
        assert mode == 0 and "ncclReduce" in note
        g.draw(3)
        img = g.gather()
        again = g.gather()
    assert np.array_equal(img, ref) and np.array_equal(again, ref)


def test_torch_distributed_calls_of_the_ranks_path_over_rccl_with_a_world_of_one():
    """bench.py's N > 1 path and distributed.py over backend nccl (= RCCL) — init with device_id, reduce, gather, all_gather, all_reduce, barrier, both assembles of a renderer's
    image — with ONE rank on the box's GPU, in a process of its own (two ranks cannot share a device under RCCL; the two-rank runs of the suite use gloo)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29583")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "r06_calls", "nccl_world1.py")], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "nccl world-1 ok" in p.stdout, (p.stdout + p.stderr)[-2000:]
