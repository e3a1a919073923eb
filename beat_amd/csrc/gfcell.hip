// gfcell.hip -- multilinear Green's-function stacking for gfx950 with the rows of a cell in registers.
//
// Same arithmetic as k_gfstack (gfstack.hip; reference beat/ffi/base.py:607-709, multilinear
// branch :663-704): per (chain, target, sample) acc = fma(G[row_k], w_k, acc) over the four corner
// rows k of the chain's (duration, start-time) cell, patches ascending -- bitwise equal.
//
// With four rows per chain the lane <-> chain kernels of gfshared.hip read four LDS operands per
// FMA group and are bound by the LDS gather (19.4 ms per 512-chain launch on config 3).  Here the
// mapping is turned around:
//   workgroup  = (512-chain group, target, 64-sample tile) = 16 wavefronts
//   wavefront  = 32 chains; lane <-> sample; accumulator of chain j = VGPR pair ACC + 2j
//   per patch  : the chains of a wavefront are visited CELL BY CELL.  The four rows of a cell are
//                read once from LDS (contiguous 512-byte reads, no bank conflicts) into registers
//                and applied to every chain of the cell: 4 x v_fma_f64 with the chain's weights as
//                SGPR operands, the accumulator selected through M0 (s_set_gpr_idx_on, DST_REL |
//                SRC2_REL).  LDS traffic drops from 4 x 512 B per chain to 4 x 512 B per cell
//                (~10 cells for 32 chains once the chains are ordered, k_gc_order).
//   rows       : every distinct row segment of the group is fetched from HBM once by LDS-DMA
//                (global_load_lds_dwordx4) into a ring of three LDS buffers, two patches ahead;
//                each wavefront requests its share (no dedicated loader wavefronts: the workgroup
//                is at the 1024-thread limit).
// Everything a wavefront does is a command stream written by k_gc_tables and read with scalar
// loads one block ahead (tools/gen_gfcell_asm.py documents the block format and generates the
// wavefront program gfcell_asm.inc: the accumulators must be a contiguous physical register
// range, so the program is register-allocated by hand).
#include <cstdlib>

#include "kernels.hpp"
#include "gfcell_asm.inc"

namespace beatamd {

constexpr int GC_CG = 512;                       // chains per group
constexpr int GC_WAVES = 16;
constexpr int GC_MAXBLK = GC_NCHAIN + 2;         // lead block + at most one batch per chain
constexpr int GC_STEP_STRIDE = GC_MAXBLK * GC_BLOCK;
constexpr int GC_TPITCH = 65 * 8;                // transposed misfit tile: row pitch in bytes
constexpr int GC_PARAM_BYTES = GC_WAVES * 128;
constexpr uint32_t GC_DEAD = 0xffffffffu;

// ---------------------------------------------------------------------------- chain order
// Chains that rupture alike choose the same cells patch after patch; a wavefront that holds
// alike chains needs fewer row reads.  Order of a group: eight bands by the start-time index at
// the first patch, inside a band by the start-time index at patch P/2 (both from the row ids of
// k_gf_tables; no fault geometry needed).  Scheduling only: results do not depend on it.
struct GcOrderArgs {
    int64_t C, T, P, S;
    const uint32_t *rowoff;   // [C,T,P,4]
    int sort;
    uint32_t *order;          // [ngroups*512]: chain id or GC_DEAD
};

__global__ void __launch_bounds__(GC_CG) k_gc_order(GcOrderArgs a)
{
    __shared__ uint32_t ka[GC_CG], kb[GC_CG];
    const int tid = threadIdx.x;
    const int64_t c = (int64_t)blockIdx.x * GC_CG + tid;
    const bool live = c < a.C;
    if (!a.sort) {
        a.order[c] = live ? (uint32_t)c : GC_DEAD;
        return;
    }
    uint32_t s0 = 0xffffu, s1 = 0xffffu;
    if (live) {
        const int64_t pm = a.P / 2;
        s0 = a.rowoff[((c * a.T) * a.P) * 4 + 3] % (uint32_t)a.S;
        s1 = a.rowoff[((c * a.T) * a.P + pm) * 4 + 3] % (uint32_t)a.S;
    }
    ka[tid] = live ? ((s0 << 10) | (uint32_t)tid) : (0xffff0000u | (uint32_t)tid);
    __syncthreads();
    const int64_t nlive = min((int64_t)GC_CG, a.C - (int64_t)blockIdx.x * GC_CG);
    int r0 = 0;
    for (int k = 0; k < GC_CG; k++) r0 += ka[k] < ka[tid];
    const uint32_t band = live ? (uint32_t)((int64_t)r0 * 8 / nlive) : 15u;
    kb[tid] = (band << 28) | (s1 << 10) | (uint32_t)tid;
    __syncthreads();
    int r1 = 0;
    for (int k = 0; k < GC_CG; k++) r1 += kb[k] < kb[tid];
    a.order[(int64_t)blockIdx.x * GC_CG + r1] = live ? (uint32_t)c : GC_DEAD;
}

// ---------------------------------------------------------------------------- command stream
struct GcTabArgs {
    int nvar, ucap;
    int64_t C, T, P, DS;          // T: targets the tables are built for (1 or all)
    int64_t nsteps, step_stride;
    const uint32_t *rowoff;       // [C,T,P,4] global row ids (k_gf_tables)
    const double *fac;            // [C,T,P,4]
    ChainVec slips[4];
    const uint32_t *order;        // [ngroups*512]
    char *stream;                 // [(g*T+t)][wavefront][step 0..nsteps][step_stride]
    uint32_t *hdr;                // [(g*T+t)][step 0..nsteps+2][wavefront][8]
    uint32_t *ucount;             // [(g*T+t)*P+p] distinct rows (statistics)
};

// one workgroup of 512 threads per (group, target, patch); thread <-> position in the group order
__global__ void __launch_bounds__(GC_CG) k_gc_tables(GcTabArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    uint32_t *flags = sm;                         // [DS] presence -> slot
    uint32_t *lst = flags + ((a.DS + 1) & ~(int64_t)1);   // [128] distinct rows in row order
    uint32_t *wsum = lst + 128;                   // [16]
    uint64_t *keys = reinterpret_cast<uint64_t *>(wsum + 16);   // [512]
    uint8_t *srt = reinterpret_cast<uint8_t *>(keys + GC_CG);   // [512] sorted position -> chain slot
    const int tid = threadIdx.x;
    const int64_t gtp = blockIdx.x;
    const int64_t p = gtp % a.P;
    const int64_t gt = gtp / a.P;
    const int64_t t = gt % a.T;
    const int64_t g = gt / a.T;
    const uint32_t cid = a.order[g * GC_CG + tid];
    const bool live = cid != GC_DEAD;
    const int64_t c = live ? (int64_t)cid : 0;
    const int64_t row0 = (t * a.P + p) * a.DS;

    for (int64_t i = tid; i < a.DS; i += GC_CG) flags[i] = 0;
    __syncthreads();
    uint32_t v[4] = {0, 0, 0, 0};
    if (live)
        for (int k = 0; k < 4; k++) {
            v[k] = a.rowoff[((c * a.T + t) * a.P + p) * 4 + k] - (uint32_t)row0;
            flags[v[k]] = 1;   // benign race: every writer stores 1
        }
    __syncthreads();
    // distinct rows in row order -> dense LDS slots
    uint32_t run = 0;
    {
        const int lane = tid & 63, wv = tid >> 6, nw = GC_CG >> 6;
        for (int64_t base = 0; base < a.DS; base += GC_CG) {
            const int64_t i = base + tid;
            const uint32_t f = (i < a.DS) ? flags[i] : 0u;
            const uint64_t m = __ballot(f != 0);
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int q = 0; q < nw; q++) {
                const uint32_t x = wsum[q];
                if (q < wv) before += x;
                total += x;
            }
            if (f) {
                const uint32_t pos = run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                flags[i] = pos;
                if (pos < 128) lst[pos] = (uint32_t)i;
            }
            run += total;
            __syncthreads();
        }
    }
    const int U = (int)run;   // <= ucap <= 105 (launcher)
    if (tid == 0) a.ucount[gtp] = run;
    // cell key of the chain: its four rows
    const uint64_t key = live ? (((uint64_t)v[0] << 36) | ((uint64_t)v[1] << 24) | ((uint64_t)v[2] << 12) | (uint64_t)v[3])
                              : ~0ull;
    keys[tid] = key;
    __syncthreads();
    const int w = tid >> 5, j = tid & 31;
    {
        int r = 0;
        for (int k = 0; k < GC_NCHAIN; k++) {
            const uint64_t kk = keys[w * GC_NCHAIN + k];
            r += (kk < key) || (kk == key && k < j);
        }
        srt[w * GC_NCHAIN + r] = (uint8_t)j;
    }
    __syncthreads();
    // from here on the thread is sorted position r of wavefront w
    const int r = j;
    const int jr = srt[w * GC_NCHAIN + r];
    const uint64_t kr = keys[w * GC_NCHAIN + jr];
    const bool live_r = kr != ~0ull;
    int rs = r;
    while (rs > 0 && keys[w * GC_NCHAIN + srt[w * GC_NCHAIN + rs - 1]] == kr) rs--;
    int re = r;
    while (re + 1 < GC_NCHAIN && keys[w * GC_NCHAIN + srt[w * GC_NCHAIN + re + 1]] == kr) re++;
    const int inrun = r - rs, q = inrun & 3;
    const bool newb = live_r && q == 0;
    const uint64_t bal = __ballot(newb);
    const uint32_t half = (uint32_t)(bal >> (32 * (w & 1)));
    const int nb = __popc(half);                                    // batches of this wavefront
    const int bidx = __popc(half & (uint32_t)((2ull << r) - 1ull)) - 1;   // batch of this position
    const int nbe = nb > 0 ? nb : 1;                                // (a wavefront without chains: one empty batch)
    const uint32_t gap = (uint32_t)(a.step_stride - (int64_t)(nbe + 1) * GC_BLOCK);
    const uint32_t cr = live_r ? a.order[g * GC_CG + w * GC_NCHAIN + jr] : 0u;
    uint32_t vr[4] = {0, 0, 0, 0};
    double fr[4] = {0, 0, 0, 0};
    if (live_r)
        for (int k = 0; k < 4; k++) {
            const int64_t e = (((int64_t)cr * a.T + t) * a.P + p) * 4 + k;
            vr[k] = a.rowoff[e] - (uint32_t)row0;
            fr[k] = a.fac[e];
        }
    for (int iv = 0; iv < a.nvar; iv++) {
        const int64_t s = p * a.nvar + iv;
        // ---- row requests of the step, dealt round robin to the wavefronts
        if (tid < GC_WAVES * 8) {
            const int ww = tid >> 3, d = tid & 7;
            uint32_t val;
            if (d == 0) val = (U > ww) ? (uint32_t)((U - ww + GC_WAVES - 1) / GC_WAVES) : 0u;
            else {
                const int idx = ww + GC_WAVES * (d - 1);
                val = (idx < U) ? (lst[idx] | (flags[lst[idx]] << 16)) : 0u;
            }
            a.hdr[((gt * (a.nsteps + 3) + s) * GC_WAVES + ww) * 8 + d] = val;
        }
        // ---- blocks
        char *base = a.stream + ((gt * GC_WAVES + w) * (a.nsteps + 1) + s) * a.step_stride;
        const uint32_t ring = (uint32_t)((s % 3) * a.ucap);
        if (live_r) {
            char *blk = base + (int64_t)(1 + bidx) * GC_BLOCK;
            const double sl = a.slips[iv].base[(int64_t)cr * a.slips[iv].stride + a.slips[iv].off + p];
            double *wq = reinterpret_cast<double *>(blk + q * 32);
            for (int k = 0; k < 4; k++) wq[k] = fr[k] * sl;     // base.py:676-679 x slip, as k_gfstack
            if (newb) {
                const int cnt = min(4, re - r + 1);
                uint32_t accb = 0;
                for (int qq = 0; qq < cnt; qq++) accb |= (uint32_t)(2 * srt[w * GC_NCHAIN + r + qq]) << (8 * qq);
                const int i = 1 + bidx;                              // block index in the step
                uint32_t *info = reinterpret_cast<uint32_t *>(blk + 128);
                info[2] = accb;
                info[3] = (uint32_t)cnt | (i == nb ? 8u : 0u) | ((i == nb - 1 ? gap : 0u) << 8);
                // the rows of this batch: LDS offsets (8-byte units) in the block before it
                uint32_t *pinfo = reinterpret_cast<uint32_t *>(blk - GC_BLOCK + 128);
                const uint32_t x0 = (ring + flags[vr[0]]) * 64u, x1 = (ring + flags[vr[1]]) * 64u,
                               x2 = (ring + flags[vr[2]]) * 64u, x3 = (ring + flags[vr[3]]) * 64u;
                pinfo[0] = x0 | (x1 << 16);
                pinfo[1] = x2 | (x3 << 16);
            }
        }
        if (r == 0) {
            uint32_t *lead = reinterpret_cast<uint32_t *>(base + 128);
            lead[2] = 0;
            lead[3] = ((nbe == 1) ? gap : 0u) << 8;
            if (nb == 0) {
                lead[0] = ring * 64u; lead[1] = ring * 64u;
                uint32_t *info = reinterpret_cast<uint32_t *>(base + GC_BLOCK + 128);
                info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 8u;
            }
        }
    }
}

// ---------------------------------------------------------------------------- stacking
struct GcArgs {
    const double *G[4];
    int nvar, ucap, ntile, mode, xcd_order;
    int64_t C, T, P, N, DS, Ttab, rows_per_target, ngroups, nsteps, step_stride;
    const char *stream;
    const uint32_t *hdr, *order;
    const double *data, *wscalar;
    double *out, *partial;
};

template <int NTH>
__global__ void __launch_bounds__(1024) k_gfstack_cell(GcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t gsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile;
    int64_t t, g;
    if (a.xcd_order) {
        // the chain groups of one (target, tile) on one XCD (see k_gfstack_dma)
        const int64_t b = blockIdx.x;
        const int64_t x = b & 7, q = b >> 3;
        g = q % a.ngroups;
        const int64_t tt = (q / a.ngroups) * 8 + x;
        if (tt >= a.T * a.ntile) return;
        tile = (int)(tt % a.ntile);
        t = tt / a.ntile;
    } else {
        tile = blockIdx.x % a.ntile;
        const int64_t gt0 = blockIdx.x / a.ntile;
        t = gt0 % a.T;
        g = gt0 / a.T;
    }
    const int64_t gt = g * a.Ttab + (a.Ttab == 1 ? 0 : t);
    const int64_t n0 = (int64_t)tile * 64;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)gsm;
    const uint32_t rb0 = lds0 + GC_PARAM_BYTES;
    if (lane == 0) {
        uint32_t *pb = gsm + wave * 32;
        auto put64 = [&](int k, uint64_t x) { pb[k] = (uint32_t)x; pb[k + 1] = (uint32_t)(x >> 32); };
        put64(GC_P_ST, (uint64_t)(uintptr_t)(a.stream + ((gt * GC_WAVES + wave) * (a.nsteps + 1)) * a.step_stride));
        put64(GC_P_HD, (uint64_t)(uintptr_t)(a.hdr + ((gt * (a.nsteps + 3)) * GC_WAVES + wave) * 8));
        put64(GC_P_GROW, (uint64_t)(uintptr_t)(a.G[0] + (t * a.rows_per_target) * a.N + n0));
        pb[GC_P_DSRB] = (uint32_t)(a.DS * a.N * 8);
        pb[GC_P_ROWB] = (uint32_t)(a.N * 8);
        pb[GC_P_RB0] = rb0;
        pb[GC_P_BUFB] = (uint32_t)(a.ucap * 512);
        pb[GC_P_NSTEP] = (uint32_t)a.nsteps;
        pb[GC_P_NLANES] = (uint32_t)min((int64_t)32, (a.N - n0 + 1) / 2);
        put64(GC_P_OUT, (uint64_t)(uintptr_t)(a.out + t * a.N + n0));
        pb[GC_P_CTN] = (uint32_t)(a.T * a.N * 8);
        pb[GC_P_MODE] = (uint32_t)a.mode;
        put64(GC_P_DATA, (uint64_t)(uintptr_t)(a.data + t * a.N + n0));
        const double wt = a.wscalar ? a.wscalar[t] : 0.0;
        put64(GC_P_W, (uint64_t)__double_as_longlong(wt));
        put64(GC_P_CID, (uint64_t)(uintptr_t)(a.order + g * GC_CG + wave * GC_NCHAIN));
        put64(GC_P_PART, (uint64_t)(uintptr_t)(a.partial + t * a.ntile + tile));
        pb[GC_P_PCS] = (uint32_t)(a.T * a.ntile * 8);
        pb[GC_P_NVALID] = (uint32_t)min((int64_t)64, a.N - n0);
        pb[GC_P_TRB] = rb0 + (uint32_t)(wave * 16 * GC_TPITCH);
    }
    __syncthreads();
    const uint32_t paddr = lds0 + (uint32_t)(wave * 128);
    if (NTH) { GC_PROGRAM_1(paddr); }
    else { GC_PROGRAM_0(paddr); }
}

bool gfstack_cell_applicable(const GfStackCall &k)
{
    const SeisLib &L = *k.libs[0];
    const char *e = getenv("BEATAMD_GS_CELL");
    if (e && atoi(e) == 0) return false;
    if (k.interp != BEATAMD_MULTILINEAR || k.nvar != 1) return false;
    const char *ek = getenv("BEATAMD_GF_KERNEL");
    if (ek && atoi(ek) == 0) return false;
    if (L.N % 2 != 0) return false;
    const int64_t DS = L.D * L.S;
    if (DS > 105 || DS < 4) return false;                      // three row buffers of D*S slots in LDS
    if (L.T * L.N * 8 >= (int64_t)1 << 32 || DS * L.N * 8 >= (int64_t)1 << 32) return false;
    const bool forced = e && atoi(e) == 1;
    if (!forced && k.C < 192) return false;                    // small batches: k_gfstack_dma groups of 64..256
    if (!forced && getenv("BEATAMD_GS_CG")) return false;      // an explicit group size asks for the k_gfstack_dma family
    return true;
}

int launch_gfstack_cell(beatamd_ctx *ctx, const GfStackCall &k, const uint32_t *rowoff, const double *fac,
                        int64_t Ttab)
{
    const SeisLib &L = *k.libs[0];
    const int64_t DS = L.D * L.S;
    const int64_t ngroups = (k.C + GC_CG - 1) / GC_CG;
    const int64_t nsteps = L.P * k.nvar;
    const int64_t GT = ngroups * Ttab;
    void *p = nullptr;

    GcOrderArgs oa;
    oa.C = k.C; oa.T = Ttab; oa.P = L.P; oa.S = L.S; oa.rowoff = rowoff;
    oa.sort = !(getenv("BEATAMD_GC_SORT") && atoi(getenv("BEATAMD_GC_SORT")) == 0);
    BA_TRY(ctx->get_scratch(SL_GC_ORDER, (size_t)ngroups * GC_CG * sizeof(uint32_t), &p));
    oa.order = (uint32_t *)p;

    GcTabArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.nvar = k.nvar; ta.ucap = (int)DS;
    ta.C = k.C; ta.T = Ttab; ta.P = L.P; ta.DS = DS;
    ta.nsteps = nsteps; ta.step_stride = GC_STEP_STRIDE;
    ta.rowoff = rowoff; ta.fac = fac;
    for (int v = 0; v < k.nvar; v++) ta.slips[v] = k.slips[v];
    ta.order = oa.order;
    BA_TRY(ctx->get_scratch(SL_GC_STREAM, (size_t)GT * GC_WAVES * (nsteps + 1) * GC_STEP_STRIDE + 256, &p));
    ta.stream = (char *)p;
    const size_t hdr_pitch = (size_t)(nsteps + 3) * GC_HDR_STRIDE;
    BA_TRY(ctx->get_scratch(SL_GC_HDR, (size_t)GT * hdr_pitch, &p));
    ta.hdr = (uint32_t *)p;
    BA_TRY(ctx->get_scratch(SL_GS_UCOUNT, (size_t)GT * L.P * sizeof(uint32_t), &p));
    ta.ucount = (uint32_t *)p;
    {
        ScopedTimer tm(ctx, "grouptables");
        hipLaunchKernelGGL(k_gc_order, dim3((unsigned)ngroups), dim3(GC_CG), 0, ctx->stream, oa);
        // the three headers behind the last step stay empty
        BA_HIP(hipMemset2DAsync((char *)ta.hdr + (size_t)nsteps * GC_HDR_STRIDE, hdr_pitch, 0,
                                (size_t)3 * GC_HDR_STRIDE, (size_t)GT, ctx->stream));
        const size_t lds = (size_t)(((DS + 1) & ~(int64_t)1) + 128 + 16) * 4 + GC_CG * 8 + GC_CG;
        hipLaunchKernelGGL(k_gc_tables, dim3((unsigned)(GT * L.P)), dim3(GC_CG), lds, ctx->stream, ta);
    }
    BA_HIP(hipGetLastError());

    GcArgs a;
    memset(&a, 0, sizeof(a));
    for (int v = 0; v < k.nvar; v++) a.G[v] = k.libs[v]->g;
    a.nvar = k.nvar; a.ucap = (int)DS;
    a.ntile = (int)((L.N + 63) / 64);
    a.mode = k.mode;
    a.C = k.C; a.T = L.T; a.P = L.P; a.N = L.N; a.DS = DS;
    a.Ttab = Ttab; a.rows_per_target = L.P * DS;
    a.ngroups = ngroups; a.nsteps = nsteps; a.step_stride = GC_STEP_STRIDE;
    a.stream = ta.stream; a.hdr = ta.hdr; a.order = oa.order;
    a.data = k.data; a.wscalar = k.wscalar; a.out = k.out;
    if (k.mode == GF_RESID_SCALAR) {
        BA_TRY(ctx->get_scratch(SL_PARTIAL, (size_t)k.C * L.T * a.ntile * sizeof(double), &p));
        a.partial = (double *)p;
    }
    int64_t nblocks = ngroups * L.T * a.ntile;
    a.xcd_order = (ngroups > 1 && !(getenv("BEATAMD_GS_ORDER") && atoi(getenv("BEATAMD_GS_ORDER")) == 0)) ? 1 : 0;
    if (a.xcd_order) nblocks = ((L.T * a.ntile + 7) / 8) * 8 * ngroups;
    BA_CHECK(nblocks < (int64_t)0x7fffffff, BEATAMD_EINVAL, "gfstack: batch too large");
    int nth = (ngroups == 1) ? 1 : 0;
    {
        const char *e = getenv("BEATAMD_GS_NTHINT");
        if (e) nth = atoi(e) != 0;
    }
    const size_t ring = std::max<size_t>((size_t)3 * DS * 512, (size_t)GC_WAVES * 16 * GC_TPITCH);
    const size_t lds = GC_PARAM_BYTES + ring;
    snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "k_gfstack_cell<%d,%d>", k.mode, nth);
    ctx->gs_ngtp = GT * L.P;
    ctx->gs_trep = L.T / Ttab;
    ctx->gs_N = L.N;
    ctx->gs_cg = GC_CG;
    {
        ScopedTimer tm(ctx, "gfstack");
        void (*kern)(GcArgs) = nth ? k_gfstack_cell<1> : k_gfstack_cell<0>;
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(1024), lds, ctx->stream, a);
    }
    BA_HIP(hipGetLastError());
    if (k.mode == GF_RESID_SCALAR) BA_TRY(launch_sum_tiles(ctx, a.partial, k.C * L.T, a.ntile, k.quad));
    return BEATAMD_OK;
}

}  // namespace beatamd
