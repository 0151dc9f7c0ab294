"""Occupancy over time of ONE persistent traversal launch (diagnostics build -DMRT_WAVE_TIMES, MRT_LIB_PATH=variants/libmrt_hip_wavetimes.so):
start and end tick (100 MHz) of every wave of the first bounce + shadow launch of a serialised pass."""
import ctypes as C, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metal_raytracing_amd as mrt
from metal_raytracing_amd._ffi import lib
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h = 1920, 1080
r = mrt.Renderer((w, h), mrt.DragonScene((w, h)), seed=1)
r.set_option("frames_in_flight", 1); r.set_option("frame_batch", batch)
for kv in sys.argv[2:]:
    k, v = kv.split("="); r.set_option(k, float(v))
r.draw(2 * batch, wait=True)
r.draw(batch, wait=True)
buf = np.zeros(16384, np.uint64)
lib.mrt_debug_wave_times.restype = C.c_int
assert lib.mrt_debug_wave_times(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(-1, 2).astype(np.int64)
it = np.zeros(32768, np.uint32)
lib.mrt_debug_wave_iters.restype = C.c_int
assert lib.mrt_debug_wave_iters(it.ctypes.data_as(C.c_void_p)) == 0
it = it.reshape(-1, 4).astype(np.int64)[t[:, 1] > 0]
le8 = it[:, 0] >> 16; it[:, 0] &= 0xFFFF
t = t[t[:, 1] > 0]
t0 = t[:, 0].min(); s = (t[:, 0] - t0) / 100.0; e = (t[:, 1] - t0) / 100.0      # microseconds
T = e.max()
print(f"batch {batch}: {len(t)} waves, launch {T:.1f} us; starts: 50% by {np.percentile(s, 50):.1f}, 99% by {np.percentile(s, 99):.1f} us; ends: 1% by {np.percentile(e, 1):.1f}, 50% by {np.percentile(e, 50):.1f}, 90% by {np.percentile(e, 90):.1f}, 99% by {np.percentile(e, 99):.1f} us")
grid = np.linspace(0, T, 21)
occ = [(int(((s <= x) & (e > x)).sum())) for x in grid]
print("resident waves at 5% steps of the launch:", occ)
last = np.argsort(-t[:, 1])[:5]
for k in last:
    dur = (t[k, 1] - t[k, 0]) / 100.0
    nd = it[k, 1] & 0xFFF; maxdt = (it[k, 1] >> 12) / 100.0
    d0 = ((it[k, 3] - (t0 & 0xFFFFFFFF)) & 0xFFFFFFFF) / 100.0 if it[k, 3] else -1
    print(f"   late wave: alive {dur:.0f} us, {it[k, 0]} iterations ({dur / max(it[k, 0], 1):.2f} us each), longest single iteration {maxdt:.1f} us, drain phase from {d0:.0f} us on: {nd} iterations with {it[k, 2] / max(nd, 1):.1f} live lanes on average, {le8[k]} of them with <= 8 live lanes")
print(f"   all waves: {it[:, 0].mean():.0f} iterations on average, {1e0 * ((t[:, 1] - t[:, 0]) / 100.0).mean() / it[:, 0].mean():.2f} us per iteration")
print(f"mean residency {np.mean(e - s) / T:.3f} of the launch time (1.0 = every wave alive from start to end)")
nd_all = it[:, 1] & 0xFFF
print(f"   drain iterations per wave: mean {nd_all.mean():.1f}, p50 {np.percentile(nd_all, 50):.0f}, p99 {np.percentile(nd_all, 99):.0f}, max {nd_all.max()}; with <= 8 live lanes: mean {le8.mean():.1f}, p99 {np.percentile(le8, 99):.0f}, max {le8.max()}")
