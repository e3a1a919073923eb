// m0cost.hip -- gfx950 probe (round 4): what does a DYNAMIC accumulator cost per chain next to the static form?
//   0  static             4 x v_fmac_f64_dpp acc_j (immediate register)
//   1  index mode         s_mov m0 + 4 x v_fmac_f64_dpp acc[M0]   (s_set_gpr_idx_on, DST_REL: k_gfstack_cell)
//   2  index mode, M0 from a lane of a VGPR (v_readlane_b32 + s_mov m0)
// (v_movrels_b32 / v_movreld_b32 do not exist on gfx9 / gfx950 -- "instruction not supported on this GPU": the index
// register in s_set_gpr_idx mode is the ONLY way to address a VGPR dynamically.)
// Every variant runs 8 "chains" per loop trip; cycles per chain from s_memtime (shader clock), per wave, for
// 1 / 2 / 3 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 m0cost.hip -o m0cost && ./m0cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMA4(ACC, Q) \
    "v_fmac_f64_dpp " ACC ", %0, %1 row_newbcast:" #Q "0 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " ACC ", %0, %1 row_newbcast:" #Q "1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " ACC ", %0, %1 row_newbcast:" #Q "2 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " ACC ", %0, %1 row_newbcast:" #Q "3 row_mask:0xf bank_mask:0xf\n\t"
// (row_newbcast takes 0..15: use lanes 0..3 for every chain)
#undef FMA4
#define FMA4(ACC) \
    "v_fmac_f64_dpp " ACC ", %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " ACC ", %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " ACC ", %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " ACC ", %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"

template <int WHICH>
__global__ void __launch_bounds__(1024) k_rate(double *out, const double *w, int iters, long long *cyc, const int *idxs)
{
    const int lane = threadIdx.x & 63;
    double wv = w[lane & 15], xv = (double)lane * 1e-3;
    int iv = idxs[lane & 15];     // accumulator offsets (even, 0..14) in the lanes of a VGPR
    // accumulators v[40:55] (8 pairs), temporaries v[56:59]
    asm volatile("v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\t"
                 "v_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"
                 "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
                 ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (WHICH == 0) {
            asm volatile(FMA4("v[40:41]") FMA4("v[42:43]") FMA4("v[44:45]") FMA4("v[46:47]")
                         FMA4("v[48:49]") FMA4("v[50:51]") FMA4("v[52:53]") FMA4("v[54:55]")
                         : : "v"(wv), "v"(xv)
                         : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        } else if (WHICH == 1) {
            asm volatile("s_mov_b32 s20, 0\n\ts_set_gpr_idx_on s20, 0x8\n\t"
                         "s_mov_b32 m0, 0x8000\n\t" FMA4("v[40:41]") "s_mov_b32 m0, 0x8006\n\t" FMA4("v[40:41]")
                         "s_mov_b32 m0, 0x8002\n\t" FMA4("v[40:41]") "s_mov_b32 m0, 0x800c\n\t" FMA4("v[40:41]")
                         "s_mov_b32 m0, 0x8004\n\t" FMA4("v[40:41]") "s_mov_b32 m0, 0x800a\n\t" FMA4("v[40:41]")
                         "s_mov_b32 m0, 0x8008\n\t" FMA4("v[40:41]") "s_mov_b32 m0, 0x800e\n\t" FMA4("v[40:41]")
                         "s_set_gpr_idx_off\n\t"
                         : : "v"(wv), "v"(xv)
                         : "s20","m0","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        } else if (WHICH == 2) {
#define RL(L) "v_readlane_b32 s20, %2, " #L "\n\ts_or_b32 m0, s20, 0x8000\n\t"
            asm volatile("s_mov_b32 s20, 0\n\ts_set_gpr_idx_on s20, 0x8\n\t"
                         RL(0) FMA4("v[40:41]") RL(1) FMA4("v[40:41]") RL(2) FMA4("v[40:41]") RL(3) FMA4("v[40:41]")
                         RL(4) FMA4("v[40:41]") RL(5) FMA4("v[40:41]") RL(6) FMA4("v[40:41]") RL(7) FMA4("v[40:41]")
                         "s_set_gpr_idx_off\n\t"
                         : : "v"(wv), "v"(xv), "v"(iv)
                         : "s20","m0","scc","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    double a0;
    asm volatile("v_mov_b32 %0, v40" : "=v"(((int *)&a0)[0]) : : "v40");
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + xv;
}

int main()
{
    double *out, *w;
    long long *cyc;
    int *idx;
    hipMalloc(&out, 1 << 22); hipMalloc(&w, 128); hipMalloc(&cyc, 8 * 1024); hipMalloc(&idx, 64);
    std::vector<double> hw(16);
    std::vector<int> hi = {0, 6, 2, 12, 4, 10, 8, 14, 0, 6, 2, 12, 4, 10, 8, 14};
    for (int i = 0; i < 16; i++) hw[i] = 1e-6 * (1.0 + i);
    hipMemcpy(w, hw.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(idx, hi.data(), 64, hipMemcpyHostToDevice);
    const int iters = 20000;
    std::vector<long long> hc(1024);
    const char *names[3] = {"static accumulators", "index mode, M0 immediate", "index mode, M0 by v_readlane"};
    for (int which = 0; which < 3; which++)
        for (int nw : {4, 8, 12, 16}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(nw * 64), 0, 0, out, w, iters, cyc, idx);
                if (which == 1) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(nw * 64), 0, 0, out, w, iters, cyc, idx);
                if (which == 2) hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(nw * 64), 0, 0, out, w, iters, cyc, idx);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(hc.data(), cyc, 8 * 256, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < 256; i++) avg += hc[i];
            avg /= 256.0;
            // per chain and wave (latency view) and per chain and SIMD (throughput view: waves/SIMD chains in that time)
            const double per_chain = avg / iters / 8.0;
            printf("%-36s waves/SIMD %d: %7.1f cycles per chain and wave, %6.1f per chain and SIMD   (%.3f ms, %.2f GHz)\n",
                   names[which], nw / 4, per_chain, per_chain / (nw / 4.0), ms, avg / (ms * 1e6));
        }
    return 0;
}
