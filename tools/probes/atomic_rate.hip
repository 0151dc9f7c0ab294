// How fast does the chip take scattered 64-bit atomicMin (one per lane, pseudo-random pixels of a 1920x1080x4 buffer)?  A measuring tool for DESIGN §6 (primary visibility by
// rasterisation would need ~5 M of them per frame).   hipcc --offload-arch=gfx950 -O2 -o /tmp/atomic_rate tools/probes/atomic_rate.hip && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_atomic(unsigned long long *buf, unsigned n_slots, unsigned per_thread, int precheck, int coherent) {
    unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned x = id * 2654435761u + 12345u;
    for (unsigned k = 0; k < per_thread; k++) {
        x = x * 1664525u + 1013904223u;
        unsigned slot = coherent ? (id * 3u + (x >> 28)) % n_slots : (x >> 4) % n_slots;      // coherent: neighbouring lanes hit neighbouring pixels (a rasteriser's pattern)
        unsigned long long v = ((unsigned long long)(x | 0x40000000u) << 32) | id;
        if (precheck && buf[slot] <= v) continue;
        atomicMin(&buf[slot], v);
    }
}
int main() {
    const unsigned n_slots = 1920 * 1080 * 4; unsigned long long *buf; hipMalloc(&buf, (size_t)n_slots * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int coherent = 0; coherent < 2; coherent++) for (int pre = 0; pre < 2; pre++) {
        hipMemset(buf, 0xFF, (size_t)n_slots * 8);
        const unsigned threads = 1 << 20, per = 16;
        hipLaunchKernelGGL(k_atomic, dim3(threads / 256), dim3(256), 0, 0, buf, n_slots, 1u, pre, coherent);
        hipMemset(buf, 0xFF, (size_t)n_slots * 8);
        hipEventRecord(a); hipLaunchKernelGGL(k_atomic, dim3(threads / 256), dim3(256), 0, 0, buf, n_slots, per, pre, coherent); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("coherent %d precheck %d: %.1f M ops in %.3f ms = %.1f G/s\n", coherent, pre, threads * (double)per / 1e6, ms, threads * (double)per / ms / 1e6);
    }
    return 0;
}
