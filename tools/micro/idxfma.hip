// micro-benchmark: VGPR-relative-indexed fp64 FMA throughput on gfx950 (feasibility of keeping the
// staged GF rows in a register array selected per chain through M0).
// build: hipcc --offload-arch=gfx950 -O3 idxfma.hip -o idxfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v16d __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// R: 32 doubles = 16 slots x 2 samples, pinned v[128:191].  acc[e][2] static.
// mode A: one s_set_gpr_idx_on/off pair per entry ; mode B: mode on for 8 entries, s_set_gpr_idx_idx
#ifndef ROWS_EVERY_STEP
#define ROWS_EVERY_STEP 0
#endif
template <int MODE>
__global__ void __launch_bounds__(512) k(const double *__restrict__ rows, const uint32_t *__restrict__ idx, const double *__restrict__ w, double *__restrict__ out,
                                         int P)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double acc[32][2];
#pragma unroll
    for (int e = 0; e < 32; e++) acc[e][0] = acc[e][1] = 0.0;
    const uint32_t *ip = idx + (size_t)(blockIdx.x * 8 + wave) * P * 32;
    const double *wp = w + (size_t)(blockIdx.x * 8 + wave) * P * 32;
    v16d R0, R1;
    for (int p = 0; p < P; p++) {
        if (ROWS_EVERY_STEP || p == 0) {
        // 16 slots x 2 samples: slot j -> R[j*2 .. j*2+1]
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const double2 a = *reinterpret_cast<const double2 *>(rows + ((size_t)(p * 16 + j) * 64 + lane) * 2);
            const double2 b = *reinterpret_cast<const double2 *>(rows + ((size_t)(p * 16 + 8 + j) * 64 + lane) * 2);
            R0[2 * j] = a.x; R0[2 * j + 1] = a.y;
            R1[2 * j] = b.x; R1[2 * j + 1] = b.y;
        }
        }
#pragma unroll
        for (int c8 = 0; c8 < 4; c8++) {
            uint32_t i8[8];
            double w8[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                i8[k] = ip[p * 32 + c8 * 8 + k];
                w8[k] = wp[p * 32 + c8 * 8 + k];
            }
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    asm volatile(
                        "s_set_gpr_idx_on %[i], 0x2\n\t"
                        "v_fma_f64 %[a0], %[w], v[128:129], %[a0]\n\t"
                        "v_fma_f64 %[a1], %[w], v[130:131], %[a1]\n\t"
                        "s_set_gpr_idx_off"
                        : [a0] "+v"(acc[c8 * 8 + k][0]), [a1] "+v"(acc[c8 * 8 + k][1])
                        : [i] "s"(i8[k]), [w] "s"(w8[k]), "{v[128:159]}"(R0), "{v[160:191]}"(R1));
                }
            } else {
#define ENT(k) "s_set_gpr_idx_idx %[i" #k "]\n\t" \
               "v_fma_f64 %[a" #k "0], %[w" #k "], v[128:129], %[a" #k "0]\n\t" \
               "v_fma_f64 %[a" #k "1], %[w" #k "], v[130:131], %[a" #k "1]\n\t"
#define ACC(k) [a##k##0] "+v"(acc[c8 * 8 + k][0]), [a##k##1] "+v"(acc[c8 * 8 + k][1])
#define INP(k) [i##k] "s"(i8[k]), [w##k] "s"(w8[k])
                asm volatile(
                    "s_set_gpr_idx_on %[i0], 0x2\n\t"
                    ENT(0) ENT(1) ENT(2) ENT(3) ENT(4) ENT(5) ENT(6) ENT(7)
                    "s_set_gpr_idx_off"
                    : ACC(0), ACC(1), ACC(2), ACC(3), ACC(4), ACC(5), ACC(6), ACC(7)
                    : INP(0), INP(1), INP(2), INP(3), INP(4), INP(5), INP(6), INP(7),
                      "{v[128:159]}"(R0), "{v[160:191]}"(R1));
            }
        }
    }
    double *o = out + (size_t)(blockIdx.x * 8 + wave) * 32 * 128;
#pragma unroll
    for (int e = 0; e < 32; e++) {
        o[e * 128 + lane * 2] = acc[e][0];
        o[e * 128 + lane * 2 + 1] = acc[e][1];
    }
}

int main(int argc, char **argv)
{
    const int P = 400, NB = 2048, NW = NB * 8;
    std::vector<double> rows((size_t)P * 16 * 128), w((size_t)NW * P * 32);
    std::vector<uint32_t> idx((size_t)NW * P * 32);
    srand(1);
    for (auto &x : rows) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto &x : w) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto &x : idx) x = (rand() % 16) * 4;   // register offset of the slot (2 doubles = 4 VGPRs)
    double *d_rows, *d_w, *d_out;
    uint32_t *d_idx;
    CK(hipMalloc(&d_rows, rows.size() * 8));
    CK(hipMalloc(&d_w, w.size() * 8));
    CK(hipMalloc(&d_idx, idx.size() * 4));
    CK(hipMalloc(&d_out, (size_t)NW * 32 * 128 * 8));
    CK(hipMemcpy(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(512), 0, 0, d_rows, d_idx, d_w, d_out, P);
            else hipLaunchKernelGGL(k<1>, dim3(NB), dim3(512), 0, 0, d_rows, d_idx, d_w, d_out, P);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double fma = (double)NW * P * 32 * 2 * 64;
            printf("mode %d rep %d: %.3f ms  %.2f TFLOP/s fp64 (vector peak 78.6)\n", mode, rep, ms,
                   2 * fma / ms / 1e9);
        }
        // check wave 5 of block 7 against the host
        std::vector<double> out((size_t)32 * 128);
        const size_t wv = 7 * 8 + 5;
        CK(hipMemcpy(out.data(), d_out + wv * 32 * 128, out.size() * 8, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int e = 0; e < 32; e++)
            for (int n = 0; n < 128; n++) {
                double a = 0;
                for (int p = 0; p < P; p++) {
                    const int slot = idx[(wv * P + p) * 32 + e] / 4;
                    const double g = rows[((size_t)((ROWS_EVERY_STEP ? p : 0) * 16 + slot) * 64 + n / 2) * 2 + (n & 1)];
                    a = __builtin_fma(w[(wv * P + p) * 32 + e], g, a);
                }
                const double d = fabs(a - out[e * 128 + n]);
                if (d > maxerr) maxerr = d;
            }
        printf("mode %d max abs err vs host fma chain: %g\n", mode, maxerr);
    }
    return 0;
}
