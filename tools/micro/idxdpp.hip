// idxdpp.hip -- gfx950 probes for k_gfstack_cell (development aid):
//  (1) does VGPR indexing (s_set_gpr_idx_on, DST_REL) apply to v_fmac_f64 with a DPP row_newbcast source?
//  (2) issue cost of that instruction, of v_readlane_b32 and of a replicated ds_read_b64
//   hipcc --offload-arch=gfx950 -O2 idxdpp.hip -o idxdpp && ./idxdpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_probe(double *out, const double *w, const double *x, int idx)
{
    // accumulators v[40:47] (4 pairs); weights: lane l holds w[l % 16]; x: lane l holds x[l]
    const int lane = threadIdx.x & 63;
    double wv = w[lane & 15], xv = x[lane];
    double a0, a1, a2, a3;
    asm volatile(
        "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"
        "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
        "s_set_gpr_idx_on %6, 0x8\n\t"                       // DST_REL only (fmac: dst is also the addend)
        "v_fmac_f64_dpp v[40:41], %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "s_set_gpr_idx_off\n\t"
        "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v42\n\tv_mov_b32 %2, v44\n\tv_mov_b32 %3, v46\n\t"
        : "=v"(((int *)&a0)[0]), "=v"(((int *)&a1)[0]), "=v"(((int *)&a2)[0]), "=v"(((int *)&a3)[0])
        : "v"(wv), "v"(xv), "s"(idx)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "m0");
    // read the high halves separately (keeps the asm simple)
    int h0, h1, h2, h3;
    asm volatile("v_mov_b32 %0, v41\n\tv_mov_b32 %1, v43\n\tv_mov_b32 %2, v45\n\tv_mov_b32 %3, v47"
                 : "=v"(h0), "=v"(h1), "=v"(h2), "=v"(h3) : : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    ((int *)&a0)[1] = h0; ((int *)&a1)[1] = h1; ((int *)&a2)[1] = h2; ((int *)&a3)[1] = h3;
    out[lane * 4 + 0] = a0; out[lane * 4 + 1] = a1; out[lane * 4 + 2] = a2; out[lane * 4 + 3] = a3;
}

// timing loops: NW waves per workgroup, one workgroup per CU
template <int WHICH>
__global__ void __launch_bounds__(1024) k_rate(double *out, const double *w, int iters, long long *cyc)
{
    __shared__ double lw[1024];
    const int lane = threadIdx.x & 63;
    lw[threadIdx.x] = w[threadIdx.x & 15];
    __syncthreads();
    double wv = w[lane & 15], xv = (double)lane;
    int xi = lane;
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lw + (lane & 15) * 8;
    long long t0 = clock64();
    int s = 0;
    for (int i = 0; i < iters; i++) {
        if (WHICH == 0) {   // 16 indexed fmac with dpp (4 "chains")
            asm volatile(
                "s_set_gpr_idx_on %2, 0x8\n\t"
                "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "s_set_gpr_idx_on %3, 0x8\n\t"
                "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[40:41], %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                "s_set_gpr_idx_on %2, 0x8\n\t"
                "v_fmac_f64_dpp v[44:45], %0, %1 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[44:45], %0, %1 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[44:45], %0, %1 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[44:45], %0, %1 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                "s_set_gpr_idx_on %3, 0x8\n\t"
                "v_fmac_f64_dpp v[44:45], %0, %1 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[44:45], %0, %1 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[44:45], %0, %1 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp v[44:45], %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                "s_set_gpr_idx_off\n\t"
                : : "v"(wv), "v"(xv), "s"(0), "s"(2)
                : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "m0");
        } else if (WHICH == 1) {   // 16 plain fma with SGPR weights (the round-3a form)
            asm volatile(
                "v_fma_f64 v[40:41], %0, s[20:21], v[40:41]\n\tv_fma_f64 v[40:41], %0, s[20:21], v[40:41]\n\t"
                "v_fma_f64 v[40:41], %0, s[20:21], v[40:41]\n\tv_fma_f64 v[40:41], %0, s[20:21], v[40:41]\n\t"
                "v_fma_f64 v[42:43], %0, s[20:21], v[42:43]\n\tv_fma_f64 v[42:43], %0, s[20:21], v[42:43]\n\t"
                "v_fma_f64 v[42:43], %0, s[20:21], v[42:43]\n\tv_fma_f64 v[42:43], %0, s[20:21], v[42:43]\n\t"
                "v_fma_f64 v[44:45], %0, s[20:21], v[44:45]\n\tv_fma_f64 v[44:45], %0, s[20:21], v[44:45]\n\t"
                "v_fma_f64 v[44:45], %0, s[20:21], v[44:45]\n\tv_fma_f64 v[44:45], %0, s[20:21], v[44:45]\n\t"
                "v_fma_f64 v[46:47], %0, s[20:21], v[46:47]\n\tv_fma_f64 v[46:47], %0, s[20:21], v[46:47]\n\t"
                "v_fma_f64 v[46:47], %0, s[20:21], v[46:47]\n\tv_fma_f64 v[46:47], %0, s[20:21], v[46:47]\n\t"
                : : "v"(xv) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s20", "s21");
        } else if (WHICH == 2) {   // 16 v_readlane_b32
            asm volatile(
                "v_readlane_b32 s20, %0, 0\n\tv_readlane_b32 s21, %0, 1\n\tv_readlane_b32 s22, %0, 2\n\tv_readlane_b32 s23, %0, 3\n\t"
                "v_readlane_b32 s20, %0, 4\n\tv_readlane_b32 s21, %0, 5\n\tv_readlane_b32 s22, %0, 6\n\tv_readlane_b32 s23, %0, 7\n\t"
                "v_readlane_b32 s20, %0, 8\n\tv_readlane_b32 s21, %0, 9\n\tv_readlane_b32 s22, %0, 10\n\tv_readlane_b32 s23, %0, 11\n\t"
                "v_readlane_b32 s20, %0, 12\n\tv_readlane_b32 s21, %0, 13\n\tv_readlane_b32 s22, %0, 14\n\tv_readlane_b32 s23, %0, 15\n\t"
                "s_add_u32 %1, %1, s23\n\t"
                : "+v"(xi), "+s"(s) : : "s20", "s21", "s22", "s23", "scc");
        } else {   // 16 replicated ds_read_b64 (16 distinct addresses per instruction)
            asm volatile(
                "ds_read_b64 v[40:41], %0\n\tds_read_b64 v[42:43], %0 offset:128\n\tds_read_b64 v[44:45], %0 offset:256\n\tds_read_b64 v[46:47], %0 offset:384\n\t"
                "ds_read_b64 v[40:41], %0 offset:512\n\tds_read_b64 v[42:43], %0 offset:640\n\tds_read_b64 v[44:45], %0 offset:768\n\tds_read_b64 v[46:47], %0 offset:896\n\t"
                "ds_read_b64 v[40:41], %0\n\tds_read_b64 v[42:43], %0 offset:128\n\tds_read_b64 v[44:45], %0 offset:256\n\tds_read_b64 v[46:47], %0 offset:384\n\t"
                "ds_read_b64 v[40:41], %0 offset:512\n\tds_read_b64 v[42:43], %0 offset:640\n\tds_read_b64 v[44:45], %0 offset:768\n\tds_read_b64 v[46:47], %0 offset:896\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                : : "v"(la) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = xv + s + xi;
}

int main()
{
    double *out, *w, *x;
    long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&w, 128); hipMalloc(&x, 512); hipMalloc(&cyc, 8 * 1024);
    std::vector<double> hw(16), hx(64), ho(256);
    for (int i = 0; i < 16; i++) hw[i] = 1.0 + i;
    for (int i = 0; i < 64; i++) hx[i] = 100.0 + i;
    hipMemcpy(w, hw.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), 512, hipMemcpyHostToDevice);
    for (int idx = 0; idx <= 6; idx += 2) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, w, x, idx);
        hipMemcpy(ho.data(), out, 2048, hipMemcpyDeviceToHost);
        // expectation if indexing applies: accumulator idx/2 = w[5] * x[lane], the others 0
        int ok = 1, plain = 1;
        for (int l = 0; l < 64; l++)
            for (int a = 0; a < 4; a++) {
                const double want = (a == idx / 2) ? 6.0 * (100.0 + l) : 0.0;
                const double noidx = (a == 0) ? 6.0 * (100.0 + l) : 0.0;
                if (ho[l * 4 + a] != want) ok = 0;
                if (ho[l * 4 + a] != noidx) plain = 0;
            }
        printf("probe idx=%d: indexed+dpp %s (unindexed result: %s)  lane0: %g %g %g %g  lane17: %g %g %g %g\n", idx,
               ok ? "OK" : "NO", plain ? "yes" : "no", ho[0], ho[1], ho[2], ho[3], ho[68], ho[69], ho[70], ho[71]);
    }
    const int iters = 20000;
    std::vector<long long> hc(1024);
    const char *names[4] = {"16 x v_fmac_f64_dpp indexed (+4 s_set_gpr_idx_on)", "16 x v_fma_f64 sgpr weight",
                            "16 x v_readlane_b32", "16 x ds_read_b64 replicated (16 addresses)"};
    for (int which = 0; which < 4; which++)
        for (int nw : {4, 8, 16}) {
            for (int rep = 0; rep < 2; rep++) {
                if (which == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(nw * 64), 0, 0, out, w, iters, cyc);
                if (which == 1) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(nw * 64), 0, 0, out, w, iters, cyc);
                if (which == 2) hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(nw * 64), 0, 0, out, w, iters, cyc);
                if (which == 3) hipLaunchKernelGGL(k_rate<3>, dim3(256), dim3(nw * 64), 0, 0, out, w, iters, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(hc.data(), cyc, 8 * 256, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < 256; i++) avg += hc[i];
            avg /= 256.0;
            // clock64 counts at a fixed 100 MHz on this family: report it raw and per instruction group
            printf("%-52s waves/CU %2d: %10.1f ticks per 16-instruction group (x %d), per SIMD-instruction %.3f ticks\n",
                   names[which], nw, avg / iters, iters, avg / iters / 16.0 / (nw / 4.0));
        }
    return 0;
}
