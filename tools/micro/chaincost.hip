// chaincost.hip -- gfx950 probe (round 4): cycles per chain of the "chain stays in its cell" path of k_gfstack_runs
// (tools/gen_gfruns_asm.py), one control instruction added at a time:
//   0  s_bfe + s_set_gpr_idx_idx + 4 x v_fmac_f64_dpp v[ACC + M0]
//   1  0 + s_bitcmp1_b32 + s_cbranch_scc1 (never taken)
//   2  1 + s_waitcnt lgkmcnt(0)                                   = the shipped path
//   3  2 without the four FMAs (control only)
//   4  4 x v_fmac_f64_dpp, static accumulator (floor)
//   6  s_add_u32 m0, d, d (M0 = 0x8000 | offset, SCC = bit 31 of d = "next chain opens a cell") + s_cbranch_scc1 +
//      s_waitcnt + 4 FMA: ONE scalar ALU instruction per chain, descriptors d in SGPRs (scalar loads)
//   7  6 with d by v_readlane_b32 per chain
//   5  0 with the index by s_lshr_b32 of a running copy (one SALU less?  no: same count) -> s_set_gpr_idx_idx only, index
//      pre-extracted (what a per-chain scalar operand would cost)
// per wave and per SIMD for 1..4 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 chaincost.hip -o chaincost && ./chaincost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMA4 \
    "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define IDX(Q) "s_bfe_u32 s21, s22, (7 << 16) | (7 * " #Q ")\n\ts_set_gpr_idx_idx s21\n\t"
#define CHK(Q) "s_bitcmp1_b32 s22, 28 + " #Q "\n\ts_cbranch_scc1 OUT%=\n\t"
#define WAIT "s_waitcnt lgkmcnt(0)\n\t"
#define CLOB "s20","s21","s22","m0","scc","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
#define PRO "s_mov_b32 s20, 0\n\ts_set_gpr_idx_on s20, 0x8\n\ts_mov_b32 s22, %2\n\t"
#define EPI "OUT%=:\n\ts_set_gpr_idx_off\n\t"

template <int WHICH>
__global__ void __launch_bounds__(1024) k_rate(double *out, const double *w, int iters, long long *cyc, int packed)
{
    const int lane = threadIdx.x & 63;
    double wv = w[lane & 15], xv = (double)(lane + 1) * 1e-3;
    const int dtab[4] = {0x4000, 0x4003, 0x4001, 0x4006};
    int dv = dtab[lane & 3];
    asm volatile("v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\t"
                 "v_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"
                 "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
                 ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (WHICH == 0) {
            asm volatile(PRO IDX(0) FMA4 IDX(1) FMA4 IDX(2) FMA4 IDX(3) FMA4 IDX(0) FMA4 IDX(1) FMA4 IDX(2) FMA4 IDX(3) FMA4 EPI
                         : : "v"(wv), "v"(xv), "s"(packed) : CLOB);
        } else if (WHICH == 1) {
            asm volatile(PRO CHK(0) IDX(0) FMA4 CHK(1) IDX(1) FMA4 CHK(2) IDX(2) FMA4 CHK(3) IDX(3) FMA4
                         CHK(0) IDX(0) FMA4 CHK(1) IDX(1) FMA4 CHK(2) IDX(2) FMA4 CHK(3) IDX(3) FMA4 EPI
                         : : "v"(wv), "v"(xv), "s"(packed) : CLOB);
        } else if (WHICH == 2) {
            asm volatile(PRO CHK(0) WAIT IDX(0) FMA4 CHK(1) WAIT IDX(1) FMA4 CHK(2) WAIT IDX(2) FMA4 CHK(3) WAIT IDX(3) FMA4
                         CHK(0) WAIT IDX(0) FMA4 CHK(1) WAIT IDX(1) FMA4 CHK(2) WAIT IDX(2) FMA4 CHK(3) WAIT IDX(3) FMA4 EPI
                         : : "v"(wv), "v"(xv), "s"(packed) : CLOB);
        } else if (WHICH == 3) {
            asm volatile(PRO CHK(0) WAIT IDX(0) CHK(1) WAIT IDX(1) CHK(2) WAIT IDX(2) CHK(3) WAIT IDX(3)
                         CHK(0) WAIT IDX(0) CHK(1) WAIT IDX(1) CHK(2) WAIT IDX(2) CHK(3) WAIT IDX(3) EPI
                         : : "v"(wv), "v"(xv), "s"(packed) : CLOB);
        } else if (WHICH == 4) {
            asm volatile(FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 : : "v"(wv), "v"(xv), "s"(packed) : CLOB);
        } else if (WHICH == 6) {
#define ONE(R) "s_add_u32 m0, s" #R ", s" #R "\n\ts_cbranch_scc1 OUT%=\n\ts_waitcnt lgkmcnt(0)\n\t" FMA4
            asm volatile("s_mov_b32 s20, 0\n\ts_set_gpr_idx_on s20, 0x8\n\t"
                         "s_mov_b32 s24, 0x4000\n\ts_mov_b32 s25, 0x4003\n\ts_mov_b32 s26, 0x4001\n\ts_mov_b32 s27, 0x4006\n\t"
                         ONE(24) ONE(25) ONE(26) ONE(27) ONE(24) ONE(25) ONE(26) ONE(27) EPI
                         : : "v"(wv), "v"(xv), "s"(packed) : CLOB, "s24", "s25", "s26", "s27");
        } else if (WHICH == 8) {
#define ONEB(R) "s_add_u32 m0, s" #R ", s" #R "\n\t" FMA4
            asm volatile("s_mov_b32 s20, 0\n\ts_set_gpr_idx_on s20, 0x8\n\t"
                         "s_mov_b32 s24, 0x4000\n\ts_mov_b32 s25, 0x4003\n\ts_mov_b32 s26, 0x4001\n\ts_mov_b32 s27, 0x4006\n\t"
                         ONEB(24) ONEB(25) ONEB(26) ONEB(27) ONEB(24) ONEB(25) ONEB(26) ONEB(27) EPI
                         : : "v"(wv), "v"(xv), "s"(packed) : CLOB, "s24", "s25", "s26", "s27");
        } else if (WHICH == 9) {
            // taken short forward branch per chain: the new-cell block inline behind the FMAs, skipped by the chains that stay
#define SKIPPED "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
#define ONET(R, L) "s_add_u32 m0, s" #R ", s" #R "\n\t" FMA4 "s_cbranch_scc0 T" #L "%=\n\t" SKIPPED "T" #L "%=:\n\t"
            asm volatile("s_mov_b32 s20, 0\n\ts_set_gpr_idx_on s20, 0x8\n\t"
                         "s_mov_b32 s24, 0x4000\n\ts_mov_b32 s25, 0x4003\n\ts_mov_b32 s26, 0x4001\n\ts_mov_b32 s27, 0x4006\n\t"
                         ONET(24, a) ONET(25, b) ONET(26, c) ONET(27, d) ONET(24, e) ONET(25, f) ONET(26, g) ONET(27, h) EPI
                         : : "v"(wv), "v"(xv), "s"(packed) : CLOB, "s24", "s25", "s26", "s27");
        } else if (WHICH == 7) {
#define ONEV(L) "v_readlane_b32 s21, %2, " #L "\n\ts_add_u32 m0, s21, s21\n\ts_cbranch_scc1 OUT%=\n\ts_waitcnt lgkmcnt(0)\n\t" FMA4
            asm volatile("s_mov_b32 s20, 0\n\ts_set_gpr_idx_on s20, 0x8\n\t"
                         ONEV(0) ONEV(1) ONEV(2) ONEV(3) ONEV(0) ONEV(1) ONEV(2) ONEV(3) EPI
                         : : "v"(wv), "v"(xv), "v"(dv) : CLOB);
        } else if (WHICH == 5) {
#define IDX1 "s_set_gpr_idx_idx s22\n\t"
            asm volatile(PRO IDX1 FMA4 IDX1 FMA4 IDX1 FMA4 IDX1 FMA4 IDX1 FMA4 IDX1 FMA4 IDX1 FMA4 IDX1 FMA4 EPI
                         : : "v"(wv), "v"(xv), "s"(packed & 0xe) : CLOB);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    // the accumulators at offsets 0, 6, 2, 12 took the same number of chains; 4, 8, 10, 14 none
    double acc[8];
#define GET(J, R0, R1) asm volatile("v_mov_b32 %0, v" #R0 "\n\tv_mov_b32 %1, v" #R1 : "=v"(((int *)&acc[J])[0]), "=v"(((int *)&acc[J])[1]))
    GET(0, 40, 41); GET(1, 42, 43); GET(2, 44, 45); GET(3, 46, 47); GET(4, 48, 49); GET(5, 50, 51); GET(6, 52, 53); GET(7, 54, 55);
    if (blockIdx.x == 0 && threadIdx.x < 8) out[threadIdx.x] = acc[threadIdx.x];
    if (blockIdx.x == 1) out[1024 + threadIdx.x] = acc[0] + xv;
}

template <int W>
static void run(const char *name, double *out, double *w, long long *cyc, int packed)
{
    const int iters = 20000;
    std::vector<long long> hc(256);
    for (int nw : {4, 8, 12, 14, 16}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rate<W>, dim3(256), dim3(nw * 64), 0, 0, out, w, iters, cyc, packed);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(hc.data(), cyc, 8 * 256, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < 256; i++) avg += hc[i];
        avg /= 256.0;
        const double per_chain = avg / iters / 8.0;
        double ha[8];
        hipMemcpy(ha, out, 64, hipMemcpyDeviceToHost);
        // wall time per chain and SIMD at 2.0 GHz (s_memtime is not the shader clock)
        printf("%-44s waves/CU %2d: %.3f ms = %5.1f ns per chain and wave, %5.2f ns per chain and SIMD | acc %.4g %.4g %.4g %.4g %.4g %.4g %.4g %.4g\n",
               name, nw, ms, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) / ((nw + 3) / 4), ha[0], ha[1], ha[2], ha[3], ha[4], ha[5], ha[6], ha[7]);
    }
}

int main()
{
    double *out, *w;
    long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&w, 128); hipMalloc(&cyc, 8 * 1024);
    std::vector<double> hw(16);
    for (int i = 0; i < 16; i++) hw[i] = 1e-6 * (1.0 + i);
    hipMemcpy(w, hw.data(), 128, hipMemcpyHostToDevice);
    const int packed = (0 << 0) | (6 << 7) | (2 << 14) | (12 << 21);   // four 7-bit offsets, no "new cell" bit
    run<4>("4 FMA, static accumulator", out, w, cyc, packed);
    run<5>("set_gpr_idx_idx + 4 FMA", out, w, cyc, packed);
    run<0>("s_bfe + set_gpr_idx_idx + 4 FMA", out, w, cyc, packed);
    run<1>("+ s_bitcmp1 + s_cbranch_scc1 (untaken)", out, w, cyc, packed);
    run<2>("+ s_waitcnt lgkmcnt(0)   (shipped path)", out, w, cyc, packed);
    run<3>("shipped path without the FMAs", out, w, cyc, packed);
    run<6>("s_add_u32 m0 (+SCC) + cbranch + wait + 4 FMA", out, w, cyc, packed);
    run<8>("s_add_u32 m0 + 4 FMA back to back", out, w, cyc, packed);
    run<9>("s_add_u32 m0 + 4 FMA + TAKEN branch over 12 instr", out, w, cyc, packed);
    run<7>("v_readlane + s_add_u32 m0 + cbranch + wait + 4 FMA", out, w, cyc, packed);
    return 0;
}
