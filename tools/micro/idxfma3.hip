// micro-benchmark 3: CW=16 chains x 2 samples/lane, 4 waves/SIMD (<=128 VGPRs), 8 row slots in v[96:127]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v16d __attribute__((ext_vector_type(16)));
typedef double d8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define FMA2(k) \
    "v_fma_f64 %[a" #k "0], %[w" #k "], v[96:97], %[a" #k "0]\n\t" \
    "v_fma_f64 %[a" #k "1], %[w" #k "], v[98:99], %[a" #k "1]\n\t"
#define ENTP(k, tcur, tnext, word, sh) \
    "s_set_gpr_idx_idx " #tcur "\n\t" "s_bfe_u32 " #tnext ", %[" #word "], " #sh "\n\t" FMA2(k)
#define ENTL(k, tcur) "s_set_gpr_idx_idx " #tcur "\n\t" FMA2(k)
#define ACC(k) [a##k##0] "+v"(acc[e0 + k][0]), [a##k##1] "+v"(acc[e0 + k][1])
#define OUTS ACC(0), ACC(1), ACC(2), ACC(3), ACC(4), ACC(5), ACC(6), ACC(7)
#define INS [w0] "s"(w[0]), [w1] "s"(w[1]), [w2] "s"(w[2]), [w3] "s"(w[3]), [w4] "s"(w[4]), [w5] "s"(w[5]), \
          [w6] "s"(w[6]), [w7] "s"(w[7]), [ia] "s"(ia), [ib] "s"(ib), "{v[96:127]}"(R0)

__device__ __forceinline__ void fma8(double (&acc)[16][2], const int e0, const d8 w, uint32_t ia, uint32_t ib,
                                     const v16d &R0)
{
    asm volatile(
        "s_bfe_u32 s100, %[ia], 0x80000\n\t"
        "s_set_gpr_idx_on s100, 0x2\n\t"
        ENTP(0, s100, s101, ia, 0x80008) ENTP(1, s101, s100, ia, 0x80010) ENTP(2, s100, s101, ia, 0x80018)
        ENTP(3, s101, s100, ib, 0x80000) ENTP(4, s100, s101, ib, 0x80008) ENTP(5, s101, s100, ib, 0x80010)
        ENTP(6, s100, s101, ib, 0x80018) ENTL(7, s101)
        "s_set_gpr_idx_off"
        : OUTS : INS : "s100", "s101", "scc");
}

template <int NOLOAD>
__global__ void __launch_bounds__(1024) k(const double *__restrict__ rows, const u4 *__restrict__ idx,
                                          const d8 *__restrict__ w, double *__restrict__ out, int P, int nshare)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double acc[16][2];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e][0] = acc[e][1] = 0.0;
    const int stream = (blockIdx.x / nshare) * 16 + wave;
    const u4 *ip = idx + (size_t)stream * P;
    const d8 *wp = w + (size_t)stream * P * 2;
    v16d R0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const double2 a = *reinterpret_cast<const double2 *>(rows + ((size_t)j * 64 + lane) * 2);
        R0[2 * j] = a.x; R0[2 * j + 1] = a.y;
    }
    u4 ic = ip[0];
    d8 wa = wp[0], wb = wp[1];
    for (int p = 0; p < P; p++) {
        asm volatile("" :: "s"(ic), "s"(wa), "s"(wb));
        const int pn = NOLOAD ? 0 : min(p + 1, P - 1);
        const u4 in_ = ip[pn];
        const d8 wan = wp[2 * pn], wbn = wp[2 * pn + 1];
        fma8(acc, 0, wa, ic[0], ic[1], R0);
        fma8(acc, 8, wb, ic[2], ic[3], R0);
        ic = in_; wa = wan; wb = wbn;
    }
    double *o = out + (size_t)(blockIdx.x * 16 + wave) * 16 * 128;
#pragma unroll
    for (int e = 0; e < 16; e++)
#pragma unroll
        for (int s = 0; s < 2; s++) o[e * 128 + lane * 2 + s] = acc[e][s];
}

int main(int argc, char **argv)
{
    const int P = 400, NB = 2048, nshare = (argc > 1 ? atoi(argv[1]) : 16), NS = NB / nshare * 16;
    std::vector<double> rows((size_t)8 * 128), w((size_t)NS * P * 16);
    std::vector<uint32_t> idx((size_t)NS * P * 4);
    srand(1);
    for (auto &x : rows) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto &x : w) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto &x : idx) {
        x = 0;
        for (int b = 0; b < 4; b++) x |= (uint32_t)((rand() % 8) * 4) << (8 * b);  // slot * 4 VGPRs
    }
    double *d_rows, *d_w, *d_out;
    uint32_t *d_idx;
    CK(hipMalloc(&d_rows, rows.size() * 8));
    CK(hipMalloc(&d_w, w.size() * 8));
    CK(hipMalloc(&d_idx, idx.size() * 4));
    CK(hipMalloc(&d_out, (size_t)NB * 16 * 16 * 128 * 8));
    CK(hipMemcpy(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int var = 0; var < 2; var++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            auto kk = var == 0 ? k<0> : k<1>;
            hipLaunchKernelGGL(kk, dim3(NB), dim3(1024), 0, 0, d_rows, (const u4 *)d_idx, (const d8 *)d_w, d_out, P, nshare);
            CK(hipGetLastError());
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double fma = (double)NB * 16 * P * 16 * 2 * 64;
            printf("var %d rep %d: %.3f ms  %.2f TFLOP/s fp64 (vector peak 78.6)\n", var, rep, ms, 2 * fma / ms / 1e9);
        }
        std::vector<double> out((size_t)16 * 128);
        const size_t blk = 37, wv = 11, st = (blk / nshare) * 16 + wv;
        CK(hipMemcpy(out.data(), d_out + (blk * 16 + wv) * 16 * 128, out.size() * 8, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int e = 0; e < 16; e++)
            for (int n = 0; n < 128; n++) {
                double a = 0;
                for (int p = 0; p < P; p++) {
                    const uint32_t word = idx[(st * P + p) * 4 + e / 4];
                    const int slot = ((word >> (8 * (e % 4))) & 255) / 4;
                    const double g = rows[((size_t)slot * 64 + n / 2) * 2 + (n & 1)];
                    a = __builtin_fma(w[(st * P + p) * 16 + e], g, a);
                }
                const double d = fabs(a - out[e * 128 + n]);
                if (d > maxerr) maxerr = d;
            }
        printf("var %d max abs err vs host fma chain: %g\n", var, maxerr);
    }
    return 0;
}
