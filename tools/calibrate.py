"""Prints the on-chip calibration (mrt_debug_calibrate) as one JSON line: VALU issue capacity and divergent-gather rates."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metal_raytracing_amd as mrt
from metal_raytracing_amd._ffi import lib, check

def calibrate(ctx, table_mb=128):
    out = (C.c_double * 5)()
    check(lib.mrt_debug_calibrate(ctx.handle, int(table_mb) << 20, out))
    return {"v_fma_f32_wave_insts_per_s": out[0], "v_pk_fma_f32_wave_insts_per_s": out[1], "shader_clock_Hz": out[4],
            "v_fma_f32_cycles_per_wave_inst_per_simd": out[4] / (out[0] / 1024.0) if out[0] else None,
            "gather16_bytes_per_s": out[2], "gather80_bytes_per_s": out[3], "table_MiB": table_mb}

if __name__ == "__main__":
    ctx = mrt.Context(0)
    res = {}
    for mb in (2, 16, 128, 1024):
        res[f"{mb}MiB"] = calibrate(ctx, mb)
    print(json.dumps(res))
