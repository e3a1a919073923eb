// micro-benchmark: the LDS gather + FMA phase of k_gfstack_dma alone (no DMA, no barriers, no tables):
// every lane reads 64 doubles of "its" row (one of U rows, pitch 65 doubles) with hand-issued
// ds_read_b64 (16 in flight) and does 64 fp64 FMAs, 400 steps, 4096 workgroups x 8 wavefronts --
// the practical ceiling of the lane <-> chain mapping on this part.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int OFF>
__device__ __forceinline__ void rd8(double (&x)[8], uint32_t addr, int tok)
{
    asm("ds_read_b64 %0, %8 offset:%c10\n\tds_read_b64 %1, %8 offset:%c10+8\n\tds_read_b64 %2, %8 offset:%c10+16\n\t"
        "ds_read_b64 %3, %8 offset:%c10+24\n\tds_read_b64 %4, %8 offset:%c10+32\n\tds_read_b64 %5, %8 offset:%c10+40\n\t"
        "ds_read_b64 %6, %8 offset:%c10+48\n\tds_read_b64 %7, %8 offset:%c10+56"
        : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7])
        : "v"(addr), "s"(tok), "n"(OFF));
}
template <int NLEFT>
__device__ __forceinline__ void wait8(double (&x)[8])
{
    asm("s_waitcnt lgkmcnt(%c8)"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "n"(NLEFT));
}
template <int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 6 ? 3 : 2, NW == 6 ? 3 : 2))) k(double *out, int nsteps, int U)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 75 * 65 * 2; i += NW * 64) lds[i] = 1e-3 * (i % 97);
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; i++) acc[i] = 0.0;
    uint32_t h = tid * 2654435761u + blockIdx.x * 40503u;
    for (int s = 0; s < nsteps; s++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t slot = (h >> 16) % (uint32_t)U;
        const double w = 1.0 + 1e-6 * (h & 1023);
        const uint32_t xs = lds0 + (uint32_t)(((s & 1) * 75 * 65 + slot * 65) * 8);
        double ya[8], yb[8];
        rd8<0>(ya, xs, s);
        rd8<64>(yb, xs, s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gq = 0; gq < 8; gq++) {
            double(&cur)[8] = (gq & 1) ? yb : ya;
            if (gq + 1 < 8) wait8<8>(cur); else wait8<0>(cur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; q++) acc[gq * 8 + q] = fma(cur[q], w, acc[gq * 8 + q]);
            __builtin_amdgcn_sched_barrier(0);
            switch (gq + 2) {
            case 2: rd8<128>(cur, xs, s); break;
            case 3: rd8<192>(cur, xs, s); break;
            case 4: rd8<256>(cur, xs, s); break;
            case 5: rd8<320>(cur, xs, s); break;
            case 6: rd8<384>(cur, xs, s); break;
            case 7: rd8<448>(cur, xs, s); break;
            default: break;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double q = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) q += acc[i];
    out[(size_t)blockIdx.x * (NW * 64) + tid] = q;
}
int main()
{
    double *d; CK(hipMalloc(&d, (size_t)8192 * 512 * 8));
    CK(hipFuncSetAttribute((const void *)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 79200));
    CK(hipFuncSetAttribute((const void *)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 79200));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nw : {8, 6}) {
        // same FMA count: 512 chains x 64 targets x 64 tiles x 400 steps = 4096 workgroups of 8 wavefronts
        const int nblocks = 4096 * 8 / nw;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (nw == 8) hipLaunchKernelGGL(k<8>, dim3(nblocks), dim3(512), 79200, 0, d, 400, 21);
            else hipLaunchKernelGGL(k<6>, dim3(nblocks), dim3(384), 79200, 0, d, 400, 21);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%d wavefronts per workgroup (%s per CU): %.3f ms per launch-equivalent\n", nw,
                            nw == 8 ? "8 waves" : "12 waves", ms);
        }
    }
    return 0;
}
