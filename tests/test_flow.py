"""Flow passes (renderer option flow = 1, csrc/flow.h): after the primary trace ONE launch carries a pass; chunks of rays are handed from wave to
wave inside it.  The image must be the wavefront pipeline's, and the oracle's, bit for bit — whatever the chunk size, the session length, the
number of waves, and however early idle waves leave."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def render(mrt, ctx, sc, size, frames, draws=1, **opts):
    r = mrt.Renderer(size, sc, ctx=ctx)
    for k, v in opts.items():
        r.set_option(k, v)
    for _ in range(draws):
        r.draw(frames, wait=True)
    img = r.accumulation().copy(); st = r.stats
    r.close()
    return img, st


@pytest.fixture(scope="module")
def small_dragon(mrt, gpu_ctx):
    sc = mrt.DragonScene((320, 200))
    ref, st0 = render(mrt, gpu_ctx, sc, (320, 200), 8, draws=2)
    r = mrt.Renderer((320, 200), sc, ctx=gpu_ctx)
    r.set_option("flow", 1)
    yield r, ref, st0
    r.close()


DEFAULTS = dict(flow_chunk=512, flow_take=8, flow_granule=128, flow_session_rays=2048, flow_mix=1, flow_order=1, flow_slots=0, flow_idle_polls=4096, flow_exit_rays=512, frame_batch=8, frames_in_flight=6)


@pytest.mark.parametrize("opts", [dict(), dict(flow_chunk=64, flow_take=1, flow_granule=64, flow_session_rays=64), dict(flow_chunk=4096, flow_take=16, flow_granule=512, flow_session_rays=65536), dict(flow_order=0), dict(flow_mix=0), dict(flow_take=2, flow_session_rays=64), dict(flow_slots=64), dict(flow_slots=1),
                                  dict(flow_idle_polls=0), dict(flow_exit_rays=0), dict(flow_exit_rays=65536), dict(frame_batch=1), dict(frame_batch=4, frames_in_flight=12), dict(frame_batch=3, frames_in_flight=2)], ids=str)
def test_flow_image_is_the_pipelines(small_dragon, opts):
    r, ref, st0 = small_dragon
    for k, v in {**DEFAULTS, **opts}.items():
        r.set_option(k, v)
    r.frameIndex = 0; r.reset_stats()
    r.draw(8, wait=True); r.draw(8, wait=True)
    st1 = r.stats
    assert np.array_equal(r.accumulation().view(np.uint32), ref.view(np.uint32))
    assert (st1.closest_rays, st1.shadow_rays, st1.primary_rays) == (st0.closest_rays, st0.shadow_rays, st0.primary_rays)


@pytest.mark.parametrize("bounces", [1, 2, 3])
def test_flow_matches_oracle(mrt, orc, gpu_ctx, bounces):
    from test_gpu_parity import assert_parity, oracle_render
    sc = mrt.CornellScene((96, 96))
    ref, _ = oracle_render(orc, mrt, sc, 96, 96, 3, bounces=bounces)
    r = mrt.Renderer((96, 96), sc, ctx=gpu_ctx, max_bounces=bounces)
    r.set_option("flow", 1)
    r.draw(3, wait=True)
    assert_parity(r.accumulation(), ref, exact_frac=1.0)
    r.close()


def test_flow_1080p_dragon_and_ragged_size(mrt, gpu_ctx):
    sc = mrt.DragonScene((1920, 1080))
    ref, st0 = render(mrt, gpu_ctx, sc, (1920, 1080), 12)
    img, st1 = render(mrt, gpu_ctx, sc, (1920, 1080), 12, flow=1)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert (st1.closest_rays, st1.shadow_rays) == (st0.closest_rays, st0.shadow_rays)
    ref, _ = render(mrt, gpu_ctx, sc, (333, 211), 5)
    img, _ = render(mrt, gpu_ctx, sc, (333, 211), 5, flow=1, frames_in_flight=3)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_flow_falls_back_where_it_does_not_apply(mrt, gpu_ctx):
    """materials, more than three bounces: the option is accepted and the wavefront pipeline runs"""
    sc = mrt.CornellScene((160, 100))
    ref, _ = render(mrt, gpu_ctx, sc, (160, 100), 4, materials=1)
    img, _ = render(mrt, gpu_ctx, sc, (160, 100), 4, materials=1, flow=1)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_shadow_planes_are_the_contribution_queue(mrt, gpu_ctx):
    """shadow_planes (default): contribution planes + one byte per shadow ray that got through; = 0: the contribution queue and the read-modify-write of the
    sample buffer.  Same additions in the same order: same image, same ray counts — flattened, two-level, sharded, one- and four-frame passes."""
    for scene, sopts, size in ((mrt.DragonScene((640, 360)), None, (640, 360)), (mrt.InstancedDragonScene((320, 200)), {"instancing": 1}, (320, 200))):
        imgs = []
        for planes in (1, 0):
            r = mrt.Renderer(size, scene, ctx=gpu_ctx, scene_options=sopts)
            r.set_option("shadow_planes", planes)
            r.draw(5, wait=True); r.set_option("frame_batch", 1); r.draw(2, wait=True)
            st = r.stats
            imgs.append((r.accumulation().copy(), (st.closest_rays, st.shadow_rays)))
            r.close()
        assert np.array_equal(imgs[0][0].view(np.uint32), imgs[1][0].view(np.uint32)) and imgs[0][1] == imgs[1][1]
    sc = mrt.DragonScene((333, 211))
    parts = []
    for planes in (1, 0):
        acc = np.zeros((211, 333, 4), np.float32)
        for rank in range(3):
            r = mrt.Renderer((333, 211), sc, ctx=gpu_ctx); r.set_option("shadow_planes", planes); r.set_shard(rank, 3); r.draw(3, wait=True)
            acc += r.accumulation(); r.close()
        parts.append(acc)
    assert np.array_equal(parts[0].view(np.uint32), parts[1].view(np.uint32))


@pytest.mark.parametrize("opts", [dict(tail_accumulate=0), dict(fuse_primary=0), dict(fuse_primary=2, frames_in_flight=1, frame_batch=1), dict(fuse_primary=0, shadow_planes=0, tail_accumulate=0),
                                  dict(frames_in_flight=2, frame_batch=3), dict(primary_hint=0),
                                  dict(stream_even=0), dict(stream_even=100, frame_batch=1), dict(stream_even=1600, frames_in_flight=2, frame_batch=2), dict(persistent=0), dict(persistent=1, persist_chunk=64)], ids=str)
def test_pipeline_switches_do_not_change_the_image(mrt, gpu_ctx, opts):
    """fuse_primary (primary rays traced inside shade(0)), tail_accumulate (the last passes folded in one launch), shadow_planes, stream_even / persistent (how a traversal launch
    hands its rays to its waves): scheduling and storage, never the image —
    over draws of several shapes (a draw shorter than the lanes, one that wraps around them, a single frame)."""
    sc = mrt.DragonScene((400, 240))
    imgs = []
    for o in (dict(), opts):
        r = mrt.Renderer((400, 240), sc, ctx=gpu_ctx)
        for k, v in o.items():
            r.set_option(k, v)
        for n in (3, 1, 17, 52):
            r.draw(n, wait=True)
        st = r.stats
        imgs.append((r.accumulation().copy(), (st.closest_rays, st.shadow_rays, st.primary_rays, st.frames)))
        r.close()
    assert np.array_equal(imgs[0][0].view(np.uint32), imgs[1][0].view(np.uint32)) and imgs[0][1] == imgs[1][1]


@pytest.mark.parametrize("bounces,size", [(1, (96, 96)), (2, (96, 96)), (3, (9, 7)), (3, (1, 1)), (2, (65, 3))])
def test_default_path_matches_oracle_at_few_bounces_and_tiny_sizes(mrt, orc, gpu_ctx, bounces, size):
    """the default path (primary rays traced inside shade(0), shadow planes, tail accumulate) against the oracle where its special cases live:
    one and two bounces (no bounce queue at the last one), images smaller than a tile, a single pixel"""
    from test_gpu_parity import assert_parity, oracle_render
    sc = mrt.CornellScene(size)
    ref, cnt = oracle_render(orc, mrt, sc, size[0], size[1], 6, bounces=bounces)
    r = mrt.Renderer(size, sc, ctx=gpu_ctx, max_bounces=bounces)
    r.draw(5, wait=True); r.draw(1, wait=True)
    assert_parity(r.accumulation(), ref, exact_frac=1.0)
    st = r.stats
    assert (st.closest_rays, st.shadow_rays) == tuple(cnt)
    r.close()
