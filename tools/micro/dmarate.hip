// micro-benchmark: LDS-DMA (global_load_lds_dwordx4) instruction rate per CU on gfx950.
//   mode 0: 32 active lanes per DMA (512 B), mode 1: 64 lanes (1 KiB)
//   W wavefronts per workgroup (one workgroup per CU: 120 KB LDS), K DMAs per wavefront and
//   iteration, vmcnt(0) + barrier per iteration (the k_gfstack_dma step structure without compute)
// build: hipcc --offload-arch=gfx950 -O3 dmarate.hip -o dmarate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int FULL>
__global__ void __launch_bounds__(512) k(const char *src, size_t span, int iters, int K, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    const bool on = FULL ? true : lane < 32;
    const unsigned voff = (unsigned)(lane * 16);
    unsigned keep = 0;
    size_t pos = ((size_t)blockIdx.x * 977 + wave * 131) * 4096 % span;
    for (int it = 0; it < iters; it++) {
        for (int k = 0; k < K; k++) {
            const char *rowp = src + pos;
            pos = (pos + 40960 * 7) % span;
            const unsigned dst = lds0 + (unsigned)(((it & 1) * nw * K + wave * K + k) * 1024);
            if (on) {
                unsigned tok;
                asm("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 %0, 0"
                    : "=s"(tok) : "v"(voff), "s"(rowp), "s"(dst));
                keep |= tok;
            }
        }
        unsigned tok;
        asm("s_waitcnt vmcnt(0)\n\ts_mov_b32 %0, 0" : "=s"(tok) : "s"(it));
        keep |= tok;
        __builtin_amdgcn_s_barrier();
    }
    if (keep) sink[0] = keep;
}

int main()
{
    const size_t big = (size_t)24 << 30, small = (size_t)48 << 20;
    char *d; unsigned *sink;
    CK(hipMalloc(&d, big)); CK(hipMemset(d, 1, big)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CK(hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    const int iters = 400, NB = 256 * 4;
    for (int full = 0; full < 2; full++)
        for (int W : {2, 4, 8})
            for (int K : {1, 3, 6})
                for (int where = 0; where < 2; where++) {
                    if ((size_t)2 * W * K * 1024 > 120 * 1024) continue;
                    const size_t span = where ? big : small;
                    float best = 1e9;
                    for (int rep = 0; rep < 3; rep++) {
                        CK(hipEventRecord(e0));
                        if (full) hipLaunchKernelGGL(k<1>, dim3(NB), dim3(W * 64), 120 * 1024, 0, d, span, iters, K, sink);
                        else hipLaunchKernelGGL(k<0>, dim3(NB), dim3(W * 64), 120 * 1024, 0, d, span, iters, K, sink);
                        CK(hipGetLastError());
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    const double ndma_cu = (double)NB / 256 * iters * W * K;   // DMA instructions per CU
                    const double bytes = (double)NB * iters * W * K * (full ? 1024 : 512);
                    printf("%s lanes W=%d K=%d %s: %.3f ms  %.0f clk/DMA/CU(2.4GHz)  %.2f TB/s chip  per-iteration %.0f clk\n",
                           full ? "64" : "32", W, K, where ? "HBM" : "L2 ", best, best * 1e-3 * 2.4e9 / ndma_cu,
                           bytes / best / 1e9, best * 1e-3 * 2.4e9 / (NB / 256 * iters));
                }
    return 0;
}
