// micro-benchmark: compute phase of the lane <-> sample mapping proposed for round 2.
//   wavefront = 32 chains x 128 samples (2 per lane); accumulators static (128 VGPRs);
//   per step: the wavefront's <= 16 distinct rows are read from LDS into a pinned VGPR array
//   (ds_read_b128, consecutive lanes -> conflict-free), each chain's (register offset, weight) comes
//   from one per-lane LDS read + v_readlane, and its 2 FMAs select the row through M0.
// Same FMA count as ldsgather.hip (512 chains x 64 targets x 400 patches x 4096 samples).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v16d __attribute__((ext_vector_type(16)));
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define FMA2(k) \
    "s_set_gpr_idx_idx %[i" #k "]\n\t" \
    "v_fma_f64 %[a" #k "0], %[w" #k "], v[128:129], %[a" #k "0]\n\t" \
    "v_fma_f64 %[a" #k "1], %[w" #k "], v[130:131], %[a" #k "1]\n\t"
#define ACC(k) [a##k##0] "+v"(acc[e0 + k][0]), [a##k##1] "+v"(acc[e0 + k][1])
#define INP(k) [i##k] "s"(idx[k]), [w##k] "s"(w[k])

__device__ __forceinline__ void fma8(double (&acc)[32][2], const int e0, const uint32_t (&idx)[8], const double (&w)[8],
                                     const v16d &R0, const v16d &R1)
{
    asm volatile(
        "s_set_gpr_idx_on %[i0], 0x2\n\t"
        FMA2(0) FMA2(1) FMA2(2) FMA2(3) FMA2(4) FMA2(5) FMA2(6) FMA2(7)
        "s_set_gpr_idx_off"
        : ACC(0), ACC(1), ACC(2), ACC(3), ACC(4), ACC(5), ACC(6), ACC(7)
        : INP(0), INP(1), INP(2), INP(3), INP(4), INP(5), INP(6), INP(7), "{v[128:159]}"(R0), "{v[160:191]}"(R1));
}

// LDS: rows[NR][128] doubles (1 KiB each), then per step-slot metadata meta[MS][8 waves][32] of {u32 regoff, pad, double w}
constexpr int NR = 32, MS = 16;
__global__ void __launch_bounds__(512) k(double *out, int nsteps, int U, int check)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *rows = lds;
    uint4 *meta = reinterpret_cast<uint4 *>(lds + NR * 128);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NR * 128; i += 512) rows[i] = 1e-3 * ((i * 7) % 101) + (i / 128);
    for (int i = tid; i < MS * 8 * 32; i += 512) {
        uint32_t h = (uint32_t)i * 2654435761u + 12345u;
        h ^= h >> 13;
        const uint32_t slot = h % (uint32_t)U;          // the wave's local row slot (U <= 16)
        const double w = 1.0 + 1e-3 * (h & 255);
        uint4 m;
        m.x = slot * 4;                                  // register offset of the slot (2 doubles = 4 VGPRs)
        m.y = 0;
        m.z = (uint32_t)(__double_as_longlong(w) & 0xffffffffu);
        m.w = (uint32_t)(__double_as_longlong(w) >> 32);
        meta[i] = m;
    }
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    double acc[32][2];
#pragma unroll
    for (int e = 0; e < 32; e++) acc[e][0] = acc[e][1] = 0.0;
    for (int s = 0; s < nsteps; s++) {
        // this lane's chain entry (lanes 0..31) of the step
        const uint4 m = meta[((s % MS) * 8 + wave) * 32 + (lane & 31)];
        // the wave's rows -> pinned registers: slot j <- row (j + s + wave) % NR   (16 B per lane)
        v16d R0, R1;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j < U) {
                const int r = (j + s + wave) % NR;
                const v2d x = *reinterpret_cast<const v2d *>(rows + r * 128 + lane * 2);
                if (j < 8) { R0[2 * j] = x.x; R0[2 * j + 1] = x.y; }
                else { R1[2 * (j - 8)] = x.x; R1[2 * (j - 8) + 1] = x.y; }
            }
        }
#pragma unroll
        for (int c8 = 0; c8 < 4; c8++) {
            uint32_t idx[8];
            double w[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int e = c8 * 8 + q;
                idx[q] = __builtin_amdgcn_readlane(m.x, e);
                const uint32_t lo = __builtin_amdgcn_readlane(m.z, e), hi = __builtin_amdgcn_readlane(m.w, e);
                w[q] = __longlong_as_double(((long long)hi << 32) | lo);
            }
            fma8(acc, c8 * 8, idx, w, R0, R1);
        }
    }
    if (check) {
        double *o = out + ((size_t)blockIdx.x * 8 + wave) * 32 * 128;
#pragma unroll
        for (int e = 0; e < 32; e++) {
            o[e * 128 + lane * 2] = acc[e][0];
            o[e * 128 + lane * 2 + 1] = acc[e][1];
        }
    } else {
        double q = 0;
#pragma unroll
        for (int e = 0; e < 32; e++) q += acc[e][0] + acc[e][1];
        out[(size_t)blockIdx.x * 512 + tid] = q;
    }
}

int main()
{
    const size_t ldsb = NR * 128 * 8 + MS * 8 * 32 * 16;
    double *d; CK(hipMalloc(&d, (size_t)4096 * 8 * 32 * 128 * 8));
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // correctness on a small grid
    {
        const int U = 11, ns = 37;
        hipLaunchKernelGGL(k, dim3(4), dim3(512), ldsb, 0, d, ns, U, 1);
        CK(hipDeviceSynchronize());
        std::vector<double> o((size_t)32 * 128);
        const int blk = 3, wv = 5;
        CK(hipMemcpy(o.data(), d + ((size_t)blk * 8 + wv) * 32 * 128, o.size() * 8, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int e = 0; e < 32; e++)
            for (int n = 0; n < 128; n++) {
                double a = 0;
                for (int s = 0; s < ns; s++) {
                    uint32_t i = (uint32_t)(((s % MS) * 8 + wv) * 32 + e);
                    uint32_t h = i * 2654435761u + 12345u; h ^= h >> 13;
                    const int slot = h % U; const double w = 1.0 + 1e-3 * (h & 255);
                    const int r = (slot + s + wv) % NR;
                    const int ii = r * 128 + n;
                    const double g = 1e-3 * ((ii * 7) % 101) + (ii / 128);
                    a = __builtin_fma(w, g, a);
                }
                const double dd = fabs(a - o[e * 128 + n]);
                if (dd > maxerr) maxerr = dd;
            }
        printf("check: max abs err vs host fma chain %g\n", maxerr);
    }
    for (int U : {4, 8, 12, 16}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(4096), dim3(512), ldsb, 0, d, 400, U, 0);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("U = %2d distinct rows per wavefront: %.3f ms per launch-equivalent\n", U, ms);
        }
    }
    return 0;
}
