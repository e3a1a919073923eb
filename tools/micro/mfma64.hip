// micro-benchmark: achievable v_mfma_f64_16x16x4_f64 rate on gfx950 (register-resident operands)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int NACC>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0)
{
    v4f64 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = v4f64{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    double *d; CK(hipMalloc(&d, 256 * 4096 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int nb : {256, 512, 1024, 2048}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<8>, dim3(nb), dim3(256), 0, 0, d, iters, 1.0, 2.0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = (double)nb * 4 * iters * 8 * 2048.0;
            if (rep) printf("blocks %d (x4 waves): %.3f ms  %.1f TFLOP/s fp64 MFMA\n", nb, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
