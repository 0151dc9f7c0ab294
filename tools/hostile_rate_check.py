"""Performance check, not a parity test (it lived in tests/test_hostile.py until round 6, where a slow box could turn the GPU suite red for no
correctness reason): the steady-state rate on HostileDragonScene (sizes over 100 : 1, 1 % slivers) must stay >= 0.6 x the rate on DragonScene
(1920x1080, 3 bounces, same box, same run).  Exit code 1 when it does not.    usage: python tools/hostile_rate_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metal_raytracing_amd as mrt
w, h = 1920, 1080
ctx = mrt.Context(0); rate = {}
for name in ("dragon", "dragon_hostile"):
    r = mrt.Renderer((w, h), mrt.SCENES[name]((w, h)), ctx=ctx)
    r.draw(48, wait=True); r.reset_stats()
    t0 = time.perf_counter(); r.draw(144, wait=True); dt = time.perf_counter() - t0
    st = r.stats
    rate[name] = (st.closest_rays + st.shadow_rays) / dt / 1e9
    r.close()
ratio = rate["dragon_hostile"] / rate["dragon"]
print(f"Grays/s: dragon {rate['dragon']:.2f}, hostile {rate['dragon_hostile']:.2f}, ratio {ratio:.3f} (bar 0.6)")
sys.exit(0 if ratio >= 0.6 else 1)
