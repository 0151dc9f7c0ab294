// valu_rates.hip — issue cost of the VALU instructions the traversal kernels are made of, measured on the chip.
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates
// Every SIMD holds 8 waves; each wave issues ITERS x 16 instances of one instruction on 16 independent registers.
// Prints, per instruction, wave64 instructions per second over the whole chip and cycles per instruction per SIMD at the
// shader clock the chip held during that loop (s_memtime against the 100 MHz s_memrealtime).  A measuring tool, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITERS = 2048;

#define KERNEL(NAME, ASM, ...)                                                                                          \
    __global__ void __launch_bounds__(64, 8) NAME(unsigned *out, unsigned long long *clocks, unsigned seed) {           \
        unsigned a[16];                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < 16; k++) a[k] = seed * (k + 3) + threadIdx.x;                             \
        unsigned b = seed | 0x3f800000u, c = (seed * 7u) | 0x3f000000u;                                                 \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();              \
        for (int i = 0; i < ITERS; i++) {                                                                               \
            _Pragma("unroll") for (int k = 0; k < 16; k++) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c) __VA_ARGS__); \
        }                                                                                                               \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();              \
        unsigned s = 0;                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < 16; k++) s ^= a[k];                                                       \
        if (s == 0x12345678u) out[blockIdx.x] = s;                                                                      \
        if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = r1 - r0; }              \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_cvt_ub0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL(k_cvt_ub2, "v_cvt_f32_ubyte2 %0, %0")
KERNEL(k_cvt_u32, "v_cvt_f32_u32 %0, %0")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", : "vcc")
KERNEL(k_cmp, "v_cmp_le_f32 vcc, %0, %1", : "vcc")
KERNEL(k_cmp_sgpr, "v_cmp_le_f32 s[20:21], %0, %1", : "s20", "s21")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 8")
KERNEL(k_bfe_i, "v_bfe_i32 %0, %0, 3, 8")
KERNEL(k_bfm, "v_bfm_b32 %0, %0, %1")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %2", : "vcc")
KERNEL(k_div_fmas, "v_div_fmas_f32 %0, %0, %1, %2", : "vcc")
KERNEL(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %2")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c")
KERNEL(k_bpermute, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")
KERNEL(k_readlane, "v_readfirstlane_b32 s20, %0", : "s20")
KERNEL(k_fma_mix_lo, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]")
KERNEL(k_fma_mix_hi, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %0, vcc", : "vcc")
KERNEL(k_or, "v_or_b32 %0, %0, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_min, "v_min_f32 %0, %0, %1")
KERNEL(k_mul_legacy, "v_mul_legacy_f32 %0, %0, %1")
KERNEL(k_or_sdwa, "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(k_add_f32_sdwa, "v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(k_mul_f32_sdwa, "v_mul_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(k_cmp_class, "v_cmp_class_f32 vcc, %0, %1", : "vcc")
KERNEL(k_subrev, "v_subrev_u32 %0, %0, %1")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
KERNEL(k_pk_min_f16, "v_pk_min_f16 %0, %0, %1")
KERNEL(k_fma_f16, "v_fma_f16 %0, %0, %1, %2")
KERNEL(k_max_f16, "v_max_f16 %0, %0, %1")
KERNEL(k_cvt_f16_f32, "v_cvt_f16_f32 %0, %0")
KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_pk_lshl_b16, "v_pk_lshlrev_b16 %0, 1, %0")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(k_dot2_f16, "v_dot2_f32_f16 %0, %1, %2, %0")
KERNEL(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %0, %1, %2")
KERNEL(k_min3_u32, "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_max_u32, "v_max_u32 %0, %0, %1")
KERNEL(k_sdwa_cvt, "v_cvt_f32_ubyte0_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")

#define KERNEL64(NAME, ASM, ...)                                                                                        \
    __global__ void __launch_bounds__(64, 8) NAME(unsigned *out, unsigned long long *clocks, unsigned seed) {           \
        unsigned long long a[8];                                                                                        \
        _Pragma("unroll") for (int k = 0; k < 8; k++) a[k] = seed * (k + 3) + threadIdx.x;                              \
        unsigned b = seed | 0x3f800000u, c = (seed * 7u) | 0x3f000000u;                                                 \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();              \
        for (int i = 0; i < ITERS; i++) {                                                                               \
            _Pragma("unroll") for (int r = 0; r < 2; r++)                                                               \
            _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c) __VA_ARGS__);  \
        }                                                                                                               \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();              \
        unsigned long long s = 0;                                                                                       \
        _Pragma("unroll") for (int k = 0; k < 8; k++) s ^= a[k];                                                        \
        if (s == 0x12345678u) out[blockIdx.x] = (unsigned)s;                                                            \
        if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = r1 - r0; }              \
    }
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0", : "vcc")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 4, %0")
KERNEL64(k_pk_fma, "v_pk_fma_f32 %0, %0, %0, %0")
KERNEL64(k_pk_mul, "v_pk_mul_f32 %0, %0, %0")
KERNEL64(k_pk_add, "v_pk_add_f32 %0, %0, %0")
KERNEL64(k_cvt_pk_fp8, "v_cvt_pk_f32_fp8 %0, %1")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %0")

typedef void (*Kern)(unsigned *, unsigned long long *, unsigned);
struct Entry { const char *name; Kern k; };

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    unsigned *out; unsigned long long *clocks;
    CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&clocks, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<Entry> es = {
#define E(n) {#n, n}
        E(k_fma), E(k_fmac), E(k_mul), E(k_add), E(k_sub), E(k_max), E(k_max3), E(k_min3), E(k_med3), E(k_cvt_ub0), E(k_cvt_ub2), E(k_sdwa_cvt), E(k_cvt_u32), E(k_cndmask),
        E(k_cmp), E(k_cmp_sgpr), E(k_and), E(k_xor), E(k_or3), E(k_bitop3), E(k_lshl), E(k_lshl_or), E(k_bfe), E(k_bfe_i), E(k_bfm), E(k_bfi), E(k_perm), E(k_ffbl), E(k_bcnt),
        E(k_mov), E(k_add_u32), E(k_mul_u24), E(k_mad_u24), E(k_mul_lo), E(k_mul_hi), E(k_rcp), E(k_rsq), E(k_sqrt), E(k_ldexp), E(k_div_scale), E(k_div_fmas),
        E(k_div_fixup), E(k_bpermute), E(k_readlane), E(k_fma_mix_lo), E(k_fma_mix_hi), E(k_cndmask_s), E(k_addc), E(k_or), E(k_and_or), E(k_lshr), E(k_min), E(k_mul_legacy),
        E(k_pk_fma_f16), E(k_pk_add_f16), E(k_pk_mul_f16), E(k_pk_max_f16), E(k_pk_min_f16), E(k_fma_f16), E(k_max_f16), E(k_cvt_f16_f32), E(k_cvt_pkrtz), E(k_pk_add_u16), E(k_pk_max_i16), E(k_pk_min_u16), E(k_pk_lshl_b16), E(k_pk_mad_u16), E(k_dot2_f16), E(k_sad_u8), E(k_cvt_pk_u8), E(k_min3_u32), E(k_max_u32),
        E(k_or_sdwa), E(k_add_f32_sdwa), E(k_mul_f32_sdwa), E(k_cmp_class), E(k_subrev), E(k_lshl_add), E(k_add3), E(k_xad),
        E(k_mad_u64_u32), E(k_lshl_add_u64), E(k_pk_fma), E(k_pk_mul), E(k_pk_add), E(k_cvt_pk_fp8), E(k_mov_b64)};
    const unsigned grid = (unsigned)simds * 8;
    printf("{\"device\": \"%s\", \"simds\": %d, \"rates\": {", prop.gcnArchName, simds);
    bool first = true;
    for (auto &e : es) {
        float best = 1e30f; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(64), 0, 0, out, clocks, 12345u + rep);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) { best = ms; CHECK(hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost)); }
        }
        const double insts = (double)grid * ITERS * 16.0;
        const double rate = insts / (best * 1e-3);
        const double clock = h[1] ? (double)h[0] / (double)h[1] * 1e8 : 0.0;
        const double cyc = rate > 0 ? clock * simds / rate : 0.0;      // SIMD cycles per wave64 instruction at the clock the loop ran at
        printf("%s\"%s\": {\"Ginst_per_s\": %.1f, \"clock_GHz\": %.3f, \"cycles_per_inst_per_simd\": %.2f}", first ? "" : ", ", e.name + 2, rate / 1e9, clock / 1e9, cyc);
        first = false;
    }
    printf("}}\n");
    return 0;
}
