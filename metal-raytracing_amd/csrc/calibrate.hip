// calibrate.hip — on-chip calibration of the ceilings bench.py prices the render kernels against.
//
// The traversal kernels are fp32 VALU + gather work, so besides the HBM roofline the two numbers that matter are
//   (1) the VALU issue capacity in wave64 instructions per second (MI355X_MICROARCH.md: SIMD-32, one wave64
//       v_fma_f32 per 2 cycles per SIMD when more than one wave is resident, 4 cycles for one wave alone), and
//   (2) the rate at which the vector memory path serves divergent 16-byte gathers from a table of the scene's size
//       (every traversal step is five such gathers per lane for a wide node, three for a triangle packet).
// Both are measured here instead of being assumed; bench.py reports them next to the frame's counters.
#include "renderer.h"

namespace mrt {
namespace {

// `waves` co-resident waves per SIMD each issue `iters` x 16 independent v_fma_f32
typedef float float2v __attribute__((ext_vector_type(2)));
template <int WAVES_PER_SIMD, bool PACKED>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) k_valu_issue(float *out, int iters, float seed, unsigned long long *clocks) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    if (PACKED) {
        float2v a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = float2v{seed + (float)k + (float)threadIdx.x, seed - (float)k};
        const float2v m = {1.0000001f, 1.0000002f}, c = {1e-7f, 2e-7f};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        }
#pragma unroll
        for (int k = 0; k < 8; k++) s += a[k].x + a[k].y;
    } else {
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = seed + (float)k + (float)threadIdx.x;
        const float m = 1.0000001f, c = 1e-7f;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));    // plain wave64 v_fma_f32 (the compiler would pack pairs into v_pk_fma_f32)
        }
#pragma unroll
        for (int k = 0; k < 16; k++) s += a[k];
    }
    if (s == 12345.678f) out[blockIdx.x] = s;      // never true: keeps the chain alive
    if (clocks && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {       // shader clock vs the constant 100 MHz clock over this wave's loop
        clocks[0] = __builtin_amdgcn_s_memtime() - c0; clocks[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// every lane walks a pseudo-random chain of records of R x 16 bytes (R = 1: bare 16-byte gathers; R = 5: the 80-byte wide node);
// two independent records are in flight per lane and the next addresses depend on the data, as in a traversal step
template <int R>
__global__ void __launch_bounds__(64, 8) k_gather(const float4 *__restrict__ table, uint32_t mask, int iters, float *out) {
    uint32_t idx = (blockIdx.x * 64u + threadIdx.x) * 2654435761u;
    float s = 0.0f;
    for (int i = 0; i < iters; i++) {
        uint32_t j[2];
#pragma unroll
        for (int k = 0; k < 2; k++) { idx = idx * 1664525u + 1013904223u; j[k] = ((idx >> 4) & mask) * (uint32_t)R; }
        float4 v[2][R];
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int r = 0; r < R; r++) v[k][r] = table[j[k] + r];
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int r = 0; r < R; r++) { s += v[k][r].x; idx ^= __float_as_uint(v[k][r].w) & 1u; }
    }
    if (s == 12345.678f) out[blockIdx.x] = s;
}

}  // namespace

// out[0] = wave64 v_fma_f32 instructions per second with 8 waves per SIMD (every SIMD full), out[1] = the same for v_pk_fma_f32,
// out[4] = shader clock (Hz) one wave saw during the v_fma_f32 loop (s_memtime against the 100 MHz s_memrealtime),
// out[2] = bytes per second of divergent 16-byte gathers from a table of about `table_bytes`, out[3] = the same for divergent
// 80-byte records (five consecutive 16-byte loads per lane: the wide node fetch)
int calibrate(hipStream_t stream, size_t table_bytes, double *out3) {
    int dev = 0; MRT_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop; MRT_HIP(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    // the gather index has 28 usable bits ((idx >> 4) & mask): 2^28 16-byte records = 4 GiB is the largest table the R = 1 pass can cover
    if (table_bytes > (size_t(4) << 30)) table_bytes = size_t(4) << 30;
    DevBuf<float> sink; MRT_HIP(sink.alloc(1 << 16));
    DevBuf<unsigned long long> clocks; MRT_HIP(clocks.alloc(2));       // before the events: nothing below may return without finish()
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) return MRT_ERR_HIP;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return MRT_ERR_HIP; }
    auto finish = [&](int rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; };
    const int iters = 4096;
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t grid = (uint32_t)(cus * 4 * 8);
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            if (hipEventRecord(e0, stream) != hipSuccess) return finish(MRT_ERR_HIP);
            if (pass == 0) hipLaunchKernelGGL((k_valu_issue<8, false>), dim3(grid), dim3(64), 0, stream, sink.p, iters, 1.0f, clocks.p);
            else hipLaunchKernelGGL((k_valu_issue<8, true>), dim3(grid), dim3(64), 0, stream, sink.p, iters, 1.0f, (unsigned long long *)nullptr);
            if (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return finish(MRT_ERR_HIP);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        out3[pass] = (double)grid * (double)iters * 16.0 / (best * 1e-3);
    }
    {
        unsigned long long h[2] = {0, 0};
        if (hipMemcpy(h, clocks.p, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return finish(MRT_ERR_HIP);
        out3[4] = h[1] ? (double)h[0] / (double)h[1] * 1e8 : 0.0;
    }
    for (int pass = 0; pass < 2; pass++) {
        const int R = pass == 0 ? 1 : 5;
        size_t n = 1; while (n * 2 * 16 * R <= table_bytes) n *= 2;      // records
        DevBuf<float4> table;
        if (table.alloc(n * R) != hipSuccess) return finish(MRT_ERR_OUT_OF_MEMORY);
        if (hipMemsetAsync(table.p, 0, n * R * 16, stream) != hipSuccess) return finish(MRT_ERR_HIP);
        const uint32_t grid = (uint32_t)(cus * 32);
        const int git = 256;
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            if (hipEventRecord(e0, stream) != hipSuccess) return finish(MRT_ERR_HIP);
            if (R == 1) hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(64), 0, stream, table.p, (uint32_t)(n - 1), git, sink.p);
            else hipLaunchKernelGGL(k_gather<5>, dim3(grid), dim3(64), 0, stream, table.p, (uint32_t)(n - 1), git, sink.p);
            if (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return finish(MRT_ERR_HIP);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        out3[2 + pass] = (double)grid * 64.0 * (double)git * 2.0 * 16.0 * R / (best * 1e-3);
    }
    if (hipGetLastError() != hipSuccess) return finish(MRT_ERR_HIP);
    return finish(MRT_OK);
}

}  // namespace mrt
