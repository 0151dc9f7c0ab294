"""Round 6, VERDICT item 2 probe (diagnostics build -DMRT_WAVE_TIMES, MRT_LIB_PATH=variants/libmrt_hip_wavetimes.so): what do the last live lanes of a draining wave still hold?
Subtree stealing — an idle lane taking the bottom entry of a straggler's stack — can only shorten the tail if those lanes HAVE parked sibling groups to give away."""
import ctypes as C, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metal_raytracing_amd as mrt
raw = C.CDLL(mrt.LIB_PATH)
w, h = 1920, 1080
for scene_name in ("dragon", "dragon_hostile"):
    for label, opts in (("one frame alone, whole image, static split", {"frames_in_flight": 1, "frame_batch": 1, "tile_groups": 1}),
                        ("one frame alone as 3 tile groups", {"frames_in_flight": 1, "frame_batch": 1}),
                        ("rank 0 of 8, one-frame pass", {"frames_in_flight": 1, "frame_batch": 1, "tile_groups": 1, "_shard": 8})):
        r = mrt.Renderer((w, h), mrt.SCENES[scene_name]((w, h)), seed=1)
        shard = opts.pop("_shard", 0)
        if shard: r.set_shard(0, shard)
        for k, v in opts.items(): r.set_option(k, float(v))
        r.draw(3, wait=True)
        buf = (C.c_ulonglong * 18)()
        assert raw.mrt_debug_drain_probe(buf, 1) == 0          # reset
        r.draw(4, wait=True)
        assert raw.mrt_debug_drain_probe(buf, 0) == 0
        it = np.zeros(32768, np.uint32); raw.mrt_debug_wave_iters(it.ctypes.data_as(C.c_void_p)); it = it.reshape(-1, 4).astype(np.int64)
        nd = it[:, 1] & 0xFFF; iters = it[:, 0] & 0xFFFF; used = iters > 0
        print(f"== {scene_name}: {label} (4 frames; bounce 0's bounce + shadow launch)")
        print(f"   waves {used.sum()}: iterations mean {iters[used].mean():.0f}, max {iters[used].max()}; drain iterations mean {nd[used].mean():.1f}, p99 {np.percentile(nd[used], 99):.0f}, max {nd[used].max()}")
        b = np.array(list(buf), np.float64).reshape(2, 9)
        for c, name in enumerate(("drain iterations with <= 16 live lanes", "drain iterations with <= 4 live lanes")):
            n = b[c, 8]
            if n == 0: print(f"   {name}: none"); continue
            hist = b[c, :6] / n
            cum = np.cumsum(hist)
            med = int(np.searchsorted(cum, 0.5))
            print(f"   {name}: {int(n)} lane-iterations; stack depth 0/1/2/3/4/>=5: " + " ".join(f"{x * 100:.0f}%" for x in hist) + f"; median depth {med}{'+' if med == 5 else ''}; hit children pending per lane {b[c, 6] / n:.2f}, triangles pending per lane {b[c, 7] / n:.2f}")
        r.close()
