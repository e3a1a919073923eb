// gfcell.hip -- multilinear Green's-function stacking for gfx950 with the rows of a cell in registers.
//
// Same arithmetic as k_gfstack (gfstack.hip; reference beat/ffi/base.py:607-709, multilinear
// branch :663-704): per (chain, target, sample) acc = fma(G[row_k], w_k, acc) over the four corner
// rows k of the chain's (duration, start-time) cell, patches ascending -- bitwise equal.
//
// With four rows per chain the lane <-> chain kernels of gfshared.hip read four LDS operands per
// FMA group and are bound by the LDS gather (19.4 ms per 512-chain launch on config 3).  Here the
// mapping is turned around:
//   workgroup  = (518-chain group, target, 64-sample tile) = 14 consumer + 2 loader wavefronts
//   consumer   = 37 chains; lane <-> sample; accumulator of chain j = VGPR pair ACC + 2j
//   per patch  : the chains of a wavefront are visited CELL BY CELL in batches of up to four.  The
//                four rows of a cell are read once from LDS (contiguous 512-byte reads, no bank
//                conflicts) into registers and applied to every chain of the batch:
//                    v_fmac_f64_dpp acc[M0], w, x_k row_newbcast:(4q+k)
//                the accumulator is selected through M0 (s_set_gpr_idx_on, DST_REL); the weight
//                is lane 4q+k of every 16-lane row of a VGPR pair that ONE coalesced load filled
//                with the batch's sixteen weights (lane l reads weight l mod 16).  LDS traffic
//                drops from 4 x 512 B per chain to 4 x 512 B per cell (~10 cells for 37 chains
//                once the chains are ordered, k_gc_order); no operand comes through scalar loads.
//   rows       : every distinct row segment of the group is fetched from HBM once by LDS-DMA
//                (global_load_lds_dwordx4, loader wavefronts) into a ring of three LDS buffers,
//                two patches ahead.
// What a wavefront does is table driven (k_gc_tables): per (wavefront, patch) the batch records
// (sixteen weights each), one descriptor per batch (accumulator indices, chain count, LDS offsets
// of the next batch's rows) read as one vector load per patch (lane <-> batch), and the loaders'
// request lists.  tools/gen_gfcell_asm.py generates the two wavefront programs (gfcell_asm.inc:
// the accumulators must be a contiguous physical register range, so they are register-allocated
// by hand); tools/gfcell_emu.py interprets them on the CPU (tests/test_gfcell_program.py).
#include <cstdlib>

#include "kernels.hpp"
#include "gfcell_asm.inc"
#include "gfml_asm.inc"
#include "gfruns_asm.inc"

namespace beatamd {

constexpr int GC_CG = GC_NCONS * GC_NCHAIN;    // chain slots per group (518)
constexpr int GC_WAVES = GC_NCONS + GC_NLOAD;
constexpr int GC_TB = 576;                      // threads of the table kernels (>= GC_CG)
constexpr int GC_TPITCH = 65 * 8;               // transposed misfit tile: row pitch in bytes
constexpr int GC_PARAM_BYTES = GC_WAVES * 128;
constexpr uint32_t GC_DEAD = 0xffffffffu;

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// ---------------------------------------------------------------------------- chain order
// Chains that rupture alike choose the same cells patch after patch; a wavefront that holds alike chains needs
// fewer row reads.  Two sort keys per chain: bands of whole wavefronts by the first key, inside a band by the
// second.  The keys are the hypocentre coordinates (strike, dip) when the caller has them (GfStackCall::order_key:
// the fused model path points them at the nucleation variables of q) -- 121 distinct cells per patch and 512 prior
// chains instead of 160 with the round-3 order (tests/order_experiment.py) -- and otherwise the start-time indices at
// the first patch and at patch P/2 (from the row ids of k_gf_tables; no fault geometry needed; 140).  Scheduling
// only: results do not depend on it.
struct GcOrderArgs {
    int64_t C, T, P, S;
    const uint32_t *rowoff;   // [C,T,P,4]
    int sort;
    ChainVec key[2];          // optional caller keys (base == nullptr: start-time indices)
    uint32_t *order;          // [ngroups*GC_CG]: chain id or GC_DEAD
    // batches of several groups: the chains of the whole batch in the order of the first key (k_gc_members); group g takes
    // members[g*GC_CG ..] -- a slice of the fault per group: fewer distinct rows to stage, fewer cells per wavefront.
    // nullptr: group g = chains g*GC_CG .. as they come
    uint32_t *members;
    double *key0;             // [C] first keys (scratch of k_gc_members)
};

__device__ __forceinline__ void gc_keys(const GcOrderArgs &a, int64_t c, double &f0, double &f1)
{
    if (a.key[0].base && a.key[1].base) {
        f0 = a.key[0].base[c * a.key[0].stride + a.key[0].off];
        f1 = a.key[1].base[c * a.key[1].stride + a.key[1].off];
        if (!(fabs(f0) <= 1.79e308)) f0 = 0.0;    // (NaN / inf proposals: any place will do, but a total order)
        if (!(fabs(f1) <= 1.79e308)) f1 = 0.0;
    } else {
        const int64_t pm = a.P / 2;
        f0 = (double)(a.rowoff[((c * a.T) * a.P) * 4 + 3] % (uint32_t)a.S);
        f1 = (double)(a.rowoff[((c * a.T) * a.P + pm) * 4 + 3] % (uint32_t)a.S);
    }
}

__global__ void __launch_bounds__(256) k_gc_key0(GcOrderArgs a)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.C) return;
    double f0, f1;
    gc_keys(a, c, f0, f1);
    a.key0[c] = f0;
}

// members[rank of chain c by (first key, c)] = c   (C <= GC_MEMBERS_MAX: C*C comparisons)
constexpr int64_t GC_MEMBERS_MAX = 4096;
__global__ void __launch_bounds__(256) k_gc_members(GcOrderArgs a)
{
    __shared__ double tile[256];
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double f = c < a.C ? a.key0[c] : 0.0;
    int64_t rank = 0;
    for (int64_t k0 = 0; k0 < a.C; k0 += 256) {
        __syncthreads();
        tile[threadIdx.x] = (k0 + threadIdx.x < a.C) ? a.key0[k0 + threadIdx.x] : 0.0;
        __syncthreads();
        const int n = (int)min((int64_t)256, a.C - k0);
        for (int k = 0; k < n; k++) rank += (tile[k] < f) || (tile[k] == f && k0 + k < c);
    }
    if (c < a.C) a.members[rank] = (uint32_t)c;
}

__global__ void __launch_bounds__(GC_TB) k_gc_order(GcOrderArgs a)
{
    __shared__ double ka[GC_CG], kb[GC_CG];
    __shared__ int bnd[GC_CG];
    const int tid = threadIdx.x;
    const int64_t pos = (int64_t)blockIdx.x * GC_CG + tid;
    const bool slot = tid < GC_CG;
    const bool live = slot && pos < a.C;
    if (!a.sort) {
        if (slot) a.order[pos] = live ? (uint32_t)pos : GC_DEAD;
        return;
    }
    const int64_t c = (live && a.members) ? (int64_t)a.members[pos] : pos;
    double f0 = 0.0, f1 = 0.0;
    if (live) gc_keys(a, c, f0, f1);
    if (slot) { ka[tid] = f0; kb[tid] = f1; }
    __syncthreads();
    const int nlive = (int)min((int64_t)GC_CG, a.C - (int64_t)blockIdx.x * GC_CG);
    // dead slots sort behind the live ones (tid >= nlive for all of them)
    int r0 = 0;
    if (live)
        for (int k = 0; k < nlive; k++) r0 += (ka[k] < f0) || (ka[k] == f0 && k < tid);
    const int nw = (nlive + GC_NCHAIN - 1) / GC_NCHAIN;
    const int nb = a.key[0].base ? 4 : 5;
    if (slot) bnd[tid] = live ? (r0 / GC_NCHAIN) * nb / nw : nb;
    __syncthreads();
    if (!slot) return;
    int r1 = tid;
    if (live) {
        const int b = bnd[tid];
        r1 = 0;
        for (int k = 0; k < nlive; k++)
            r1 += (bnd[k] < b) || (bnd[k] == b && ((kb[k] < f1) || (kb[k] == f1 && k < tid)));
    }
    a.order[(int64_t)blockIdx.x * GC_CG + r1] = live ? (uint32_t)c : GC_DEAD;
}

// launches the chain order of a batch into oa.order (scratch for members / keys behind it)
static int launch_gc_order(beatamd_ctx *ctx, GcOrderArgs &oa, int64_t ngroups)
{
    void *p = nullptr;
    const size_t norder = (size_t)(ngroups * GC_CG + 64);
    const bool global = oa.sort && ngroups > 1 && oa.C <= GC_MEMBERS_MAX && env_int("BEATAMD_GC_GLOBAL", 1) != 0;
    BA_TRY(ctx->get_scratch(SL_GC_ORDER, norder * sizeof(uint32_t) + (global ? (size_t)oa.C * 12 + 64 : 0), &p));
    oa.order = (uint32_t *)p;
    oa.members = nullptr;
    oa.key0 = nullptr;
    if (global) {
        oa.key0 = reinterpret_cast<double *>(((uintptr_t)(oa.order + norder) + 7) & ~(uintptr_t)7);
        oa.members = reinterpret_cast<uint32_t *>(oa.key0 + oa.C);
        const unsigned nb = (unsigned)((oa.C + 255) / 256);
        hipLaunchKernelGGL(k_gc_key0, dim3(nb), dim3(256), 0, ctx->stream, oa);
        hipLaunchKernelGGL(k_gc_members, dim3(nb), dim3(256), 0, ctx->stream, oa);
    }
    hipLaunchKernelGGL(k_gc_order, dim3((unsigned)ngroups), dim3(GC_TB), 0, ctx->stream, oa);
    return BEATAMD_OK;
}

// ---------------------------------------------------------------------------- tables
struct GcTabArgs {
    int nvar, ucap;
    int64_t C, T, P, DS;          // T: targets the tables are built for (1 or all)
    int64_t nsteps;
    const uint32_t *rowoff;       // [C,T,P,4] global row ids (k_gf_tables)
    const double *fac;            // [C,T,P,4]
    ChainVec slips[4];
    const uint32_t *order;        // [ngroups*GC_CG]
    char *wtab;                   // [(g*T+t)][consumer][step 0..nsteps][GC_NQMAX] quads of four GC_QREC-byte
                                  // records: 16 weights at +0, 16 descriptor dwords (GC_A_*) at +128
    uint32_t *ltab;               // [(g*T+t)][step 0..nsteps+2][loader][32 dwords]: count, row-pair requests
    uint32_t *ucount;             // [(g*T+t)*P+p] distinct rows (statistics)
};

// one workgroup per (group, target, patch); thread <-> chain slot of the group order
__global__ void __launch_bounds__(GC_TB) k_gc_tables(GcTabArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    uint32_t *flags = sm;                                     // [DS] presence -> slot
    uint32_t *lst = flags + ((a.DS + 1) & ~(int64_t)1);       // [128] distinct rows in row order
    uint32_t *wsum = lst + 128;                               // [16]
    uint64_t *keys = reinterpret_cast<uint64_t *>(wsum + 16); // [GC_CG]
    uint8_t *srt = reinterpret_cast<uint8_t *>(keys + GC_CG); // [GC_CG] sorted position -> chain slot of the wavefront
    uint8_t *bof = srt + GC_CG;                               // [GC_CG] sorted position -> batch index
    uint8_t *nbw = bof + GC_CG;                               // [GC_NCONS] batches per wavefront
    const int tid = threadIdx.x;
    const int64_t gtp = blockIdx.x;
    const int64_t p = gtp % a.P;
    const int64_t gt = gtp / a.P;
    const int64_t t = gt % a.T;
    const int64_t g = gt / a.T;
    const bool slot = tid < GC_CG;
    const uint32_t cid = slot ? a.order[g * GC_CG + tid] : GC_DEAD;
    const bool live = cid != GC_DEAD;
    const int64_t c = live ? (int64_t)cid : 0;
    const int64_t row0 = (t * a.P + p) * a.DS;

    for (int64_t i = tid; i < a.DS; i += GC_TB) flags[i] = 0;
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
        const int lane = tid & 63, wv = tid >> 6, nw = GC_TB >> 6;
        for (int64_t base = 0; base < a.DS; base += GC_TB) {
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
    const int U = (int)run;   // <= D*S <= 2 * GC_LREQ (launcher)
    if (tid == 0) a.ucount[gtp] = run;
    // cell key of the chain: its four rows
    const uint64_t key = live ? (((uint64_t)v[0] << 36) | ((uint64_t)v[1] << 24) | ((uint64_t)v[2] << 12) | (uint64_t)v[3])
                              : ~0ull;
    if (slot) keys[tid] = key;
    __syncthreads();
    const int w = tid / GC_NCHAIN, j = tid % GC_NCHAIN;
    if (slot) {
        int r = 0;
        for (int k = 0; k < GC_NCHAIN; k++) {
            const uint64_t kk = keys[w * GC_NCHAIN + k];
            r += (kk < key) || (kk == key && k < j);
        }
        srt[w * GC_NCHAIN + r] = (uint8_t)j;
    }
    __syncthreads();
    // batches of a wavefront: runs of equal keys in sorted order, cut after four chains (one thread per wavefront)
    if (slot && j == 0) {
        int b = -1, inrun = 0;
        uint64_t prev = ~0ull;
        for (int r = 0; r < GC_NCHAIN; r++) {
            const uint64_t kr = keys[w * GC_NCHAIN + srt[w * GC_NCHAIN + r]];
            if (kr == ~0ull) { bof[w * GC_NCHAIN + r] = 255; continue; }
            if (b < 0 || kr != prev || inrun == 4) { b++; inrun = 0; }
            prev = kr;
            inrun++;
            bof[w * GC_NCHAIN + r] = (uint8_t)b;
        }
        nbw[w] = (uint8_t)(b + 1);
    }
    __syncthreads();
    // from here on the thread is sorted position r of consumer wavefront w
    const int r = j;
    const int jr = slot ? srt[w * GC_NCHAIN + r] : 0;
    const int b = slot ? bof[w * GC_NCHAIN + r] : 255;
    const bool live_r = slot && b != 255;
    const bool first = live_r && (r == 0 || bof[w * GC_NCHAIN + r - 1] != b);
    const int q = live_r ? (first ? 0 : (r >= 1 && bof[w * GC_NCHAIN + r - 1] == b) + (r >= 2 && bof[w * GC_NCHAIN + r - 2] == b) +
                                            (r >= 3 && bof[w * GC_NCHAIN + r - 3] == b)) : 0;
    const int nb = slot ? nbw[w] : 0;
    // whole quads, at least GC_NQMIN of them (empty records pad the step)
    const int nbe = max(((nb + 3) / 4) * 4, 4 * GC_NQMIN);
    const int nq = nbe / 4;
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
        const uint32_t ring = (uint32_t)((s % 3) * a.ucap);
        // ---- row requests of the step: pairs of rows with adjacent LDS slots, dealt round robin to the
        // loader wavefronts (rowA | (rowB - rowA) << 8 | slotA << 16 | single << 24)
        for (int i = tid; i < GC_NLOAD * 32; i += GC_TB) {
            const int ll = i / 32, d = i % 32;
            const int npair = (U + 1) / 2;
            uint32_t val;
            if (d == 0) val = (npair > ll) ? (uint32_t)((npair - ll + GC_NLOAD - 1) / GC_NLOAD) : 0u;
            else {
                const int pr = ll + GC_NLOAD * (d - 1);
                if (pr < npair) {
                    const uint32_t ra = lst[2 * pr];
                    const bool single = 2 * pr + 1 >= U;
                    const uint32_t rb = single ? ra : lst[2 * pr + 1];
                    val = ra | ((rb - ra) << 8) | ((uint32_t)(2 * pr) << 16) | (single ? 1u << 24 : 0u);
                } else val = 0u;
            }
            a.ltab[((gt * (a.nsteps + 3) + s) * GC_NLOAD + ll) * 32 + d] = val;
        }
        if (!slot) continue;
        char *wb = a.wtab + ((gt * GC_NCONS + w) * (a.nsteps + 1) + s) * (int64_t)GC_WSTRIDE;
        auto rec = [&](int bb) { return wb + (int64_t)(bb >> 2) * GC_QUAD + (bb & 3) * GC_QREC; };
        auto aux = [&](int bb) { return reinterpret_cast<uint32_t *>(rec(bb) + 128); };
        auto cflags = [&](int bb) {
            return (bb == nbe - 1 ? 1u << GC_CF_LAST : 0u) | (((bb & 3) == 0 && (bb >> 2) + 2 == nq - 1) ? 1u << GC_CF_CROSS : 0u);
        };
        const uint32_t xe = ring * 512u;                        // slot 0 of the step's row buffer
        if (live_r) {
            const double sl = a.slips[iv].base[(int64_t)cr * a.slips[iv].stride + a.slips[iv].off + p];
            double *wq = reinterpret_cast<double *>(rec(b) + q * 32);
            for (int k = 0; k < 4; k++) wq[k] = fr[k] * sl;     // base.py:676-679 x slip, as k_gfstack
            if (first) {
                int cnt = 1;
                while (cnt < 4 && r + cnt < GC_NCHAIN && bof[w * GC_NCHAIN + r + cnt] == b) cnt++;
                uint32_t m[4] = {0x8000u, 0x8000u, 0x8000u, 0x8000u};   // M0 words: DST_REL | accumulator offset
                for (int qq = 0; qq < cnt; qq++) m[qq] = 0x8000u | (uint32_t)(2 * srt[w * GC_NCHAIN + r + qq]);
                uint32_t *me = aux(b);
                me[GC_A_ACC01] = m[0] | (m[1] << 16);
                me[GC_A_ACC23] = m[2] | (m[3] << 16);
                me[GC_A_CF] = (uint32_t)cnt | cflags(b);
                // LDS byte offsets of the batch's rows: its own descriptor (first batch of a step) and
                // the descriptor of the batch before it
                for (int k = 0; k < 4; k++) {
                    const uint32_t x = (ring + flags[vr[k]]) * 512u;
                    me[GC_A_XO + k] = x;
                    if (b >= 1) aux(b - 1)[GC_A_XN + k] = x;
                }
            }
        }
        // padding batches and the rows the last batch names as "next"
        if (r == 0) {
            for (int e = nb; e < nbe; e++) {
                uint32_t *me = aux(e);
                me[GC_A_ACC01] = 0x80008000u;
                me[GC_A_ACC23] = 0x80008000u;
                me[GC_A_CF] = cflags(e);
                for (int k = 0; k < 4; k++) {
                    me[GC_A_XO + k] = xe;
                    if (e >= 1) aux(e - 1)[GC_A_XN + k] = xe;
                }
            }
            for (int k = 0; k < 4; k++) aux(nbe - 1)[GC_A_XN + k] = xe;
        }
    }
}

// ---------------------------------------------------------------------------- stacking
struct GcArgs {
    const double *G[4];
    int nvar, ucap, ntile, mode, xcd_order;
    int64_t C, T, P, N, DS, Ttab, rows_per_target, ngroups, nsteps;
    const char *wtab;
    const uint32_t *ltab, *order;
    const double *data, *wscalar;
    double *out, *partial;
    const uint32_t *dtab;         // k_gfstack_runs: chain descriptors, [(g*T+t)][consumer][step 0..nsteps][GR_DLINE]
};

// VAR > 0: timing experiments of tools/gen_gfcell_asm.py (GC_ABLATIONS builds only; wrong results)
template <int NTH, int VAR>
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
        if (wave < GC_NCONS) {
            put64(GC_P_WP, (uint64_t)(uintptr_t)(a.wtab + ((gt * GC_NCONS + wave) * (a.nsteps + 1)) * (int64_t)GC_WSTRIDE));
            pb[GC_P_RB0] = rb0;
            pb[GC_P_NSTEP] = (uint32_t)a.nsteps;
            pb[GC_P_BNC] = rb0 + (uint32_t)(3 * a.ucap * 512 + wave * GC_BOUNCE);
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
        } else {
            const int ll = wave - GC_NCONS;
            put64(GC_PL_LT, (uint64_t)(uintptr_t)(a.ltab + ((gt * (a.nsteps + 3)) * GC_NLOAD + ll) * 32));
            put64(GC_PL_GROW, (uint64_t)(uintptr_t)(a.G[0] + (t * a.rows_per_target) * a.N + n0));
            pb[GC_PL_DSRB] = (uint32_t)(a.DS * a.N * 8);
            pb[GC_PL_ROWB] = (uint32_t)(a.N * 8);
            pb[GC_PL_RB0] = rb0;
            pb[GC_PL_BUFB] = (uint32_t)(a.ucap * 512);
            pb[GC_PL_NSTEP] = (uint32_t)a.nsteps;
            pb[GC_PL_NLANES] = (uint32_t)min((int64_t)32, (a.N - n0 + 1) / 2);
            // steps cycle through the slip variables' libraries patch by patch (one variable: the same base thrice)
            pb[GC_PL_NVAR] = (uint32_t)a.nvar;
            put64(GC_PL_G1, (uint64_t)(uintptr_t)(a.G[a.nvar > 1 ? 1 : 0] + (t * a.rows_per_target) * a.N + n0));
            put64(GC_PL_G2, (uint64_t)(uintptr_t)(a.G[a.nvar > 2 ? 2 : 0] + (t * a.rows_per_target) * a.N + n0));
        }
    }
    __syncthreads();
    const uint32_t paddr = lds0 + (uint32_t)(wave * 128);
    if constexpr (VAR == 0) {
        if (wave < GC_NCONS) { GC_CONSUMER_0(paddr); }
        else if (NTH) { GC_LOADER_0_1(paddr); }
        else { GC_LOADER_0_0(paddr); }
    }
#if GC_NVARIANT > 2
    if constexpr (VAR == 1) { if (wave < GC_NCONS) { GC_CONSUMER_1(paddr); } else { GC_LOADER_1_1(paddr); } }
    if constexpr (VAR == 2) { if (wave < GC_NCONS) { GC_CONSUMER_2(paddr); } else { GC_LOADER_2_1(paddr); } }
#endif
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
    // three row buffers + the wavefronts' record buffers in LDS; request table of the two loaders
    if (DS < 4 || DS > 4 * GC_LPAIR || GC_PARAM_BYTES + 3 * DS * 512 + GC_NCONS * GC_BOUNCE > 160 * 1024) return false;
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
    if (env_int("BEATAMD_GC_KEYS", 1)) { oa.key[0] = k.order_key[0]; oa.key[1] = k.order_key[1]; }
    BA_TRY(launch_gc_order(ctx, oa, ngroups));   // (reads the row ids of k_gf_tables, launched before this call)

    GcTabArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.nvar = k.nvar; ta.ucap = (int)DS;
    ta.C = k.C; ta.T = Ttab; ta.P = L.P; ta.DS = DS;
    ta.nsteps = nsteps;
    ta.rowoff = rowoff; ta.fac = fac;
    for (int v = 0; v < k.nvar; v++) ta.slips[v] = k.slips[v];
    ta.order = oa.order;
    BA_TRY(ctx->get_scratch(SL_GC_STREAM, (size_t)GT * GC_NCONS * (nsteps + 1) * GC_WSTRIDE + 4096, &p));
    ta.wtab = (char *)p;
    const size_t lt_pitch = (size_t)(nsteps + 3) * GC_NLOAD * GC_LTAB;
    BA_TRY(ctx->get_scratch(SL_GC_HDR, (size_t)GT * lt_pitch, &p));
    ta.ltab = (uint32_t *)p;
    BA_TRY(ctx->get_scratch(SL_GS_UCOUNT, (size_t)GT * L.P * sizeof(uint32_t), &p));
    ta.ucount = (uint32_t *)p;
    {
        ScopedTimer tm(ctx, "grouptables");
        // the request tables behind the last step stay empty
        BA_HIP(hipMemset2DAsync((char *)ta.ltab + (size_t)nsteps * GC_NLOAD * GC_LTAB, lt_pitch, 0,
                                (size_t)3 * GC_NLOAD * GC_LTAB, (size_t)GT, ctx->stream));
        const size_t lds = (size_t)(((DS + 1) & ~(int64_t)1) + 128 + 16) * 4 + GC_CG * 8 + GC_CG * 2 + 64;
        hipLaunchKernelGGL(k_gc_tables, dim3((unsigned)(GT * L.P)), dim3(GC_TB), lds, ctx->stream, ta);
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
    a.ngroups = ngroups; a.nsteps = nsteps;
    a.wtab = ta.wtab; a.ltab = ta.ltab; a.order = oa.order; a.dtab = nullptr;
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
    const size_t ring = std::max<size_t>((size_t)3 * DS * 512 + GC_NCONS * GC_BOUNCE, (size_t)GC_NCONS * 16 * GC_TPITCH);
    const size_t lds = GC_PARAM_BYTES + ring;
    BA_CHECK(lds <= 160 * 1024, BEATAMD_EINVAL, "internal: k_gfstack_cell row buffers exceed LDS");
    snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "k_gfstack_cell<%d,%d>", k.mode, nth);
    ctx->gs_ngtp = GT * L.P;
    ctx->gs_trep = L.T / Ttab;
    ctx->gs_N = L.N;
    ctx->gs_cg = GC_CG;
    {
        ScopedTimer tm(ctx, "gfstack");
        void (*kern)(GcArgs) = nth ? k_gfstack_cell<1, 0> : k_gfstack_cell<0, 0>;
#if GC_NVARIANT > 2
        {
            const char *ev = getenv("BEATAMD_GC_VAR");
            const int var = ev ? atoi(ev) : 0;
            if (var == 1) kern = k_gfstack_cell<1, 1>;
            if (var == 2) kern = k_gfstack_cell<1, 2>;
        }
#endif
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(1024), lds, ctx->stream, a);
    }
    BA_HIP(hipGetLastError());
    if (k.mode == GF_RESID_SCALAR) BA_TRY(launch_sum_tiles(ctx, a.partial, k.C * L.T, a.ntile, k.quad));
    return BEATAMD_OK;
}

// =============================================================================== k_gfstack_ml (round 4)
// Multilinear stacking with STATIC accumulators (tools/gen_gfml_asm.py).  Same workgroup shape, loaders,
// row ring and epilogues as k_gfstack_cell, but a consumer wavefront walks its 37 chains in a fixed order:
// no VGPR index register, no per-chain scalar work.  What makes that possible is the DENSE row layout of a
// step's LDS buffer -- slot(d, s') = d*(S+1) + s', s' = s + 1; slot s' = 0 of a duration line holds a copy of
// its LAST start-time node when a chain's floor node wrapped to it (base.py:513-517: ceil - 1 = -1 -> S-1) --
// so the four rows of ANY chain are A, A+512 (floor-duration line) and B, B+512 (ceil-duration line).
struct GmTabArgs {
    int64_t C, T, P, D, S, DS, nslot;   // T: targets the tables are built for (1 or all)
    int64_t nsteps;
    int nvar;                     // slip variables: step = patch * nvar + variable (same rows, the variable's slips)
    const uint32_t *rowoff;       // [C,T,P,4] global row ids (k_gf_tables: cc, fc, cf, ff)
    const double *fac;            // [C,T,P,4]
    ChainVec slips[4];
    const uint32_t *order;        // [ngroups*GC_CG]
    char *wtab;                   // [(g*T+t)][consumer][step 0..nsteps][GM_NREC] records of 16 entries {weight, dword, pad}
                                  // (RUNS: GR_PAIR-byte record pairs of 2 x 16 weights)
    uint32_t *ltab;               // [(g*T+t)][step 0..nsteps+2][loader][32 dwords]: count, row requests
    uint32_t *ucount;             // [(g*T+t)*P+p] row segments the loaders move (statistics)
    uint32_t *dtab;               // RUNS: [(g*T+t)][consumer][step 0..nsteps][GR_DLINE] chain descriptors (scalar loads)
};

// one workgroup per (group, target, patch); thread <-> chain slot of the group order.
// RUNS (k_gfstack_runs): the chains of a wavefront are written in CELL ORDER; records hold the weights only (GR_PAIR
// bytes per eight chains) and the descriptor line of the (wavefront, step) two dwords per sorted position:
// GR_D_BASE | chain slot | "the next position opens a new cell" << 31 (the program reads rows only then), and the
// LDS slots of the chain's row pairs, A | B << 16 (tools/gen_gfruns_asm.py).
template <int RUNS>
__global__ void __launch_bounds__(GC_TB) k_gm_tables(GmTabArgs a)
{
    __shared__ uint32_t flags[256];
    __shared__ uint32_t reqs[256];
    __shared__ uint32_t nreq_s;
    __shared__ uint32_t keys[RUNS ? GC_CG : 1];
    __shared__ uint8_t srt[RUNS ? GC_CG : 1];
    __shared__ uint8_t opens_at[RUNS ? GC_CG : 1];   // sorted position -> the chain there opens a new cell
    const int tid = threadIdx.x;
    const int64_t gtp = blockIdx.x;
    const int64_t p = gtp % a.P;
    const int64_t gt = gtp / a.P;
    const int64_t t = gt % a.T;
    const int64_t g = gt / a.T;
    const bool slot = tid < GC_CG;
    const uint32_t cid = slot ? a.order[g * GC_CG + tid] : GC_DEAD;
    const bool live = cid != GC_DEAD;
    const int64_t c = live ? (int64_t)cid : 0;
    const int64_t row0 = (t * a.P + p) * a.DS;
    const uint32_t S = (uint32_t)a.S, S1 = S + 1;

    for (int i = tid; i < 256; i += GC_TB) flags[i] = 0;
    __syncthreads();
    uint32_t sa = 0, sb = 0;       // LDS slots of (floor d, floor s) and (ceil d, floor s)
    double fr[4] = {0, 0, 0, 0};
    if (live) {
        const int64_t e = ((c * a.T + t) * a.P + p) * 4;
        const uint32_t v0 = a.rowoff[e] - (uint32_t)row0;        // (ceil d, ceil s)
        const uint32_t v2 = a.rowoff[e + 2] - (uint32_t)row0;    // (floor d, ceil s)
        const uint32_t dc = v0 / S, sc = v0 % S, df = v2 / S;
        sb = dc * S1 + sc;        // the floor node of ceil node sc is slot sc of the line (sc = 0: the wrap copy)
        sa = df * S1 + sc;
        flags[sa] = 1; flags[sa + 1] = 1; flags[sb] = 1; flags[sb + 1] = 1;   // benign race: every writer stores 1
        for (int k = 0; k < 4; k++) fr[k] = a.fac[e + k];
    }
    __syncthreads();
    // row requests in slot order: a pair moves two rows into adjacent slots (lanes 0-31 / 32-63 of one LDS-DMA
    // instruction); rowA | (rowB - rowA) << 8 | slotA << 16 | single << 24, rows relative to the step's first row
    if (tid == 0) {
        uint32_t n = 0, moved = 0;
        auto row_of = [&](uint32_t sl) { const uint32_t d = sl / S1, s1 = sl % S1; return d * S + (s1 ? s1 - 1 : S - 1); };
        for (uint32_t sl = 0; sl < (uint32_t)a.nslot;) {
            if (!flags[sl]) { sl++; continue; }
            const uint32_t ra = row_of(sl);
            if (sl + 1 < (uint32_t)a.nslot && flags[sl + 1]) {
                const uint32_t rb = row_of(sl + 1);
                if (rb > ra && rb - ra < 256) {
                    reqs[n++] = ra | ((rb - ra) << 8) | (sl << 16);
                    sl += 2; moved += 2;
                    continue;
                }
            }
            reqs[n++] = ra | (sl << 16) | (1u << 24);
            sl++; moved++;
        }
        nreq_s = n;
        a.ucount[gtp] = moved;
    }
    __syncthreads();
    const int nreq = (int)nreq_s;   // <= 2 * GC_LPAIR (launcher: (nslot + 1) / 2 + D)
    const int w = tid / GC_NCHAIN, j = tid % GC_NCHAIN;
    // cell order of the wavefront's chains (RUNS): the same for every slip variable of the patch
    uint32_t key = 0;
    int r = j;
    bool opens = false;
    if constexpr (RUNS) {
        key = live ? ((sb << 16) | sa) : 0xffffffffu;
        if (slot) keys[tid] = key;
        __syncthreads();
        if (slot) {
            r = 0;
            for (int k = 0; k < GC_NCHAIN; k++) {
                const uint32_t kk = keys[w * GC_NCHAIN + k];
                r += (kk < key) || (kk == key && k < j);
            }
            srt[w * GC_NCHAIN + r] = (uint8_t)j;
        }
        __syncthreads();
        if (slot) {
            // a chain opens a cell when its key differs from its predecessor's; dead slots ride on the rows in place
            opens = live && (r == 0 || keys[w * GC_NCHAIN + srt[w * GC_NCHAIN + r - 1]] != key);
            opens_at[w * GC_NCHAIN + r] = opens ? 1 : 0;
        }
        __syncthreads();
    }
    for (int iv = 0; iv < a.nvar; iv++) {
        const int64_t s = p * a.nvar + iv;
        for (int i = tid; i < GC_NLOAD * 32; i += GC_TB) {
            const int ll = i / 32, d = i % 32;
            uint32_t val;
            if (d == 0) val = (nreq > ll) ? (uint32_t)((nreq - ll + GC_NLOAD - 1) / GC_NLOAD) : 0u;
            else {
                const int rq = ll + GC_NLOAD * (d - 1);
                val = rq < nreq ? reqs[rq] : 0u;
            }
            a.ltab[((gt * (a.nsteps + 3) + s) * GC_NLOAD + ll) * 32 + d] = val;
        }
        if (!slot) continue;
        const uint32_t ring = (uint32_t)((s % 3) * a.nslot);
        const double sl = live ? a.slips[iv].base[c * a.slips[iv].stride + a.slips[iv].off + p] : 0.0;
        // a dead chain slot reads slot 0 of the buffer (its accumulator is never stored)
        const int q = r & 3;
        if constexpr (RUNS) {
            // weights only: a 256-byte record pair serves eight chains, entry e = {weight e of record 2p, of record 2p + 1};
            // slots and accumulator in the descriptor line
            char *rec = a.wtab + ((gt * GC_NCONS + w) * (a.nsteps + 1) + s) * (int64_t)GR_WSTRIDE + (r >> 3) * GR_PAIR +
                        ((r >> 2) & 1) * 8;
            for (int k = 0; k < 4; k++)
                *reinterpret_cast<double *>(rec + (4 * q + k) * 16) = fr[k] * sl;     // base.py:676-679 x slip, as k_gfstack
            const uint32_t next_opens = (r + 1 < GC_NCHAIN) ? (uint32_t)opens_at[w * GC_NCHAIN + r + 1] : 0u;
            uint32_t *dl = a.dtab + ((gt * GC_NCONS + w) * (a.nsteps + 1) + s) * GR_DLINE +
                           (r < GR_NHALF ? 2 * r : GR_DHALF + 2 * (r - GR_NHALF));
            dl[0] = (uint32_t)GR_D_BASE | (uint32_t)j | (next_opens << 31);
            dl[1] = (ring + sa) | ((ring + sb) << 16);
        } else {
            char *rec = a.wtab + ((gt * GC_NCONS + w) * (a.nsteps + 1) + s) * (int64_t)GM_WSTRIDE + (r >> 2) * GM_REC;
            for (int k = 0; k < 4; k++)
                *reinterpret_cast<double *>(rec + (4 * q + k) * 16) = fr[k] * sl;     // base.py:676-679 x slip, as k_gfstack
            *reinterpret_cast<uint32_t *>(rec + (2 * q) * 16 + 8) = (ring + sa) * 512u;
            *reinterpret_cast<uint32_t *>(rec + (2 * q + 1) * 16 + 8) = (ring + sb) * 512u;
        }
    }
}

template <int NTH, int VAR, int PROG>
__global__ void __launch_bounds__(1024) k_gfstack_mlr(GcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t gsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile;
    int64_t t, g;
    if (a.xcd_order) {
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
        if (wave < GC_NCONS) {
            put64(GC_P_WP, (uint64_t)(uintptr_t)(a.wtab + ((gt * GC_NCONS + wave) * (a.nsteps + 1)) *
                                                 (int64_t)(PROG == 1 ? GR_WSTRIDE : GM_WSTRIDE)));
            put64(GC_P_DP, (uint64_t)(uintptr_t)(a.dtab + ((gt * GC_NCONS + wave) * (a.nsteps + 1)) * (int64_t)GR_DLINE));
            pb[GC_P_RB0] = rb0;
            pb[GC_P_NSTEP] = (uint32_t)a.nsteps;
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
        } else {
            const int ll = wave - GC_NCONS;
            put64(GC_PL_LT, (uint64_t)(uintptr_t)(a.ltab + ((gt * (a.nsteps + 3)) * GC_NLOAD + ll) * 32));
            put64(GC_PL_GROW, (uint64_t)(uintptr_t)(a.G[0] + (t * a.rows_per_target) * a.N + n0));
            pb[GC_PL_DSRB] = (uint32_t)(a.DS * a.N * 8);
            pb[GC_PL_ROWB] = (uint32_t)(a.N * 8);
            pb[GC_PL_RB0] = rb0;
            pb[GC_PL_BUFB] = (uint32_t)(a.ucap * 512);      // ucap: slots of a step's row buffer
            pb[GC_PL_NSTEP] = (uint32_t)a.nsteps;
            pb[GC_PL_NLANES] = (uint32_t)min((int64_t)32, (a.N - n0 + 1) / 2);
            // steps cycle through the slip variables' libraries patch by patch (one variable: the same base thrice)
            pb[GC_PL_NVAR] = (uint32_t)a.nvar;
            put64(GC_PL_G1, (uint64_t)(uintptr_t)(a.G[a.nvar > 1 ? 1 : 0] + (t * a.rows_per_target) * a.N + n0));
            put64(GC_PL_G2, (uint64_t)(uintptr_t)(a.G[a.nvar > 2 ? 2 : 0] + (t * a.rows_per_target) * a.N + n0));
        }
    }
    __syncthreads();
    const uint32_t paddr = lds0 + (uint32_t)(wave * 128);
    if constexpr (PROG == 1) {
        if constexpr (VAR == 0) {
            if (wave < GC_NCONS) { GR_CONSUMER_0(paddr); }
            else if (NTH) { GC_LOADER_0_1(paddr); }
            else { GC_LOADER_0_0(paddr); }
        }
#if GR_NVARIANT > 1
        if constexpr (VAR == 1) { if (wave < GC_NCONS) { GR_CONSUMER_1(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 2) { if (wave < GC_NCONS) { GR_CONSUMER_2(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 3) { if (wave < GC_NCONS) { GR_CONSUMER_3(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 4) { if (wave < GC_NCONS) { GR_CONSUMER_4(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 5) { if (wave < GC_NCONS) { GR_CONSUMER_5(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 6) { if (wave < GC_NCONS) { GR_CONSUMER_6(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 7) { if (wave < GC_NCONS) { GR_CONSUMER_7(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 8) { if (wave < GC_NCONS) { GR_CONSUMER_8(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 9) { if (wave < GC_NCONS) { GR_CONSUMER_9(paddr); } else { GC_LOADER_0_1(paddr); } }
        if constexpr (VAR == 10) { if (wave < GC_NCONS) { GR_CONSUMER_10(paddr); } else { GC_LOADER_0_1(paddr); } }
#endif
        return;
    }
    if constexpr (VAR == 0) {
        if (wave < GC_NCONS) { GM_CONSUMER_0(paddr); }
        else if (NTH) { GC_LOADER_0_1(paddr); }
        else { GC_LOADER_0_0(paddr); }
    }
#if GM_NVARIANT > 1
#define GM_VARIANT(V, CONS, LOAD) \
    if constexpr (VAR == V) { if (wave < GC_NCONS) { CONS(paddr); } else { LOAD(paddr); } }
    GM_VARIANT(1, GM_CONSUMER_1, GC_LOADER_0_1)
    GM_VARIANT(2, GM_CONSUMER_2, GC_LOADER_0_1)
    GM_VARIANT(3, GM_CONSUMER_3, GC_LOADER_0_1)
    GM_VARIANT(4, GM_CONSUMER_4, GC_LOADER_0_1)
    GM_VARIANT(5, GM_CONSUMER_5, GM_LOADER_NODMA)
    GM_VARIANT(6, GM_CONSUMER_6, GC_LOADER_0_1)
    GM_VARIANT(7, GM_CONSUMER_7, GC_LOADER_0_1)
    GM_VARIANT(8, GM_CONSUMER_8, GM_LOADER_NODMA)
    GM_VARIANT(9, GM_CONSUMER_9, GM_LOADER_NODMA)
#endif
}


bool gfstack_ml_applicable(const GfStackCall &k)
{
    // A/B and test knobs, read per call so that one process can compare kernels (a few getenv calls next to a
    // launch of milliseconds; the launch-bound geometry step reads its knobs once)
    const int knob = env_int("BEATAMD_GS_ML", -1);   // 0: off, 1: forced also for small batches
    const int gfk = env_int("BEATAMD_GF_KERNEL", -1);
    const bool cg_fixed = getenv("BEATAMD_GS_CG") != nullptr;
    const SeisLib &L = *k.libs[0];
    if (knob == 0 || gfk == 0) return false;
    if (k.interp != BEATAMD_MULTILINEAR || k.nvar < 1 || k.nvar > 3) return false;
    if (L.N % 2 != 0) return false;
    const int64_t DS = L.D * L.S, nslot = L.D * (L.S + 1);
    // three dense row buffers in LDS; request table of the two loaders; 8-bit row / slot fields of a request
    if (DS < 1 || DS > 255 || nslot > 255 || (nslot + 1) / 2 + L.D > 2 * GC_LPAIR) return false;
    if (GC_PARAM_BYTES + 3 * nslot * 512 > 160 * 1024) return false;
    if (L.T * L.N * 8 >= (int64_t)1 << 32 || DS * L.N * 8 >= (int64_t)1 << 32) return false;
    const bool forced = knob == 1;
    if (!forced && k.C < 192) return false;          // small batches: k_gfstack_dma groups of 64..256
    if (!forced && cg_fixed) return false;           // an explicit group size asks for the k_gfstack_dma family
    return true;
}

int launch_gfstack_ml(beatamd_ctx *ctx, const GfStackCall &k, const uint32_t *rowoff, const double *fac, int64_t Ttab)
{
    const SeisLib &L = *k.libs[0];
    const int64_t DS = L.D * L.S, nslot = L.D * (L.S + 1);
    const int64_t ngroups = (k.C + GC_CG - 1) / GC_CG;
    const int64_t nsteps = L.P * k.nvar;
    const int64_t GT = ngroups * Ttab;
    void *p = nullptr;

    const bool runs = env_int("BEATAMD_GS_RUNS", 1) != 0;   // k_gfstack_runs (default) or k_gfstack_ml
    GcOrderArgs oa;
    oa.C = k.C; oa.T = Ttab; oa.P = L.P; oa.S = L.S; oa.rowoff = rowoff;
    // k_gfstack_runs: chains that rupture alike share cells patch after patch -> put them into one wavefront
    // (k_gc_order); the static program does not care
    oa.sort = runs ? env_int("BEATAMD_GC_SORT", 1) != 0 : 0;
    if (env_int("BEATAMD_GC_KEYS", 1)) { oa.key[0] = k.order_key[0]; oa.key[1] = k.order_key[1]; }
    BA_TRY(launch_gc_order(ctx, oa, ngroups));   // (reads the row ids of k_gf_tables, launched before this call)

    GmTabArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.C = k.C; ta.T = Ttab; ta.P = L.P; ta.D = L.D; ta.S = L.S; ta.DS = DS; ta.nslot = nslot;
    ta.nsteps = nsteps;
    ta.nvar = k.nvar;
    ta.rowoff = rowoff; ta.fac = fac;
    for (int v = 0; v < k.nvar; v++) ta.slips[v] = k.slips[v];
    ta.order = oa.order;
    BA_TRY(ctx->get_scratch(SL_GC_STREAM, (size_t)GT * GC_NCONS * (nsteps + 1) * GM_WSTRIDE + 8192, &p));
    ta.wtab = (char *)p;
    const size_t lt_pitch = (size_t)(nsteps + 3) * GC_NLOAD * GC_LTAB;
    BA_TRY(ctx->get_scratch(SL_GC_HDR, (size_t)GT * lt_pitch, &p));
    ta.ltab = (uint32_t *)p;
    BA_TRY(ctx->get_scratch(SL_GS_UCOUNT, (size_t)GT * L.P * sizeof(uint32_t), &p));
    ta.ucount = (uint32_t *)p;
    // (the line behind the last step is read ahead, never used)
    BA_TRY(ctx->get_scratch(SL_GC_META, (size_t)GT * GC_NCONS * (nsteps + 1) * GR_DLINE * sizeof(uint32_t) + 256, &p));
    ta.dtab = (uint32_t *)p;
    {
        ScopedTimer tm(ctx, "grouptables");
        // the request tables behind the last step stay empty
        BA_HIP(hipMemset2DAsync((char *)ta.ltab + (size_t)nsteps * GC_NLOAD * GC_LTAB, lt_pitch, 0,
                                (size_t)3 * GC_NLOAD * GC_LTAB, (size_t)GT, ctx->stream));
        if (runs) hipLaunchKernelGGL(k_gm_tables<1>, dim3((unsigned)(GT * L.P)), dim3(GC_TB), 0, ctx->stream, ta);
        else hipLaunchKernelGGL(k_gm_tables<0>, dim3((unsigned)(GT * L.P)), dim3(GC_TB), 0, ctx->stream, ta);
    }
    BA_HIP(hipGetLastError());

    GcArgs a;
    memset(&a, 0, sizeof(a));
    for (int v = 0; v < k.nvar; v++) a.G[v] = k.libs[v]->g;
    a.nvar = k.nvar; a.ucap = (int)nslot;
    a.ntile = (int)((L.N + 63) / 64);
    a.mode = k.mode;
    a.C = k.C; a.T = L.T; a.P = L.P; a.N = L.N; a.DS = DS;
    a.Ttab = Ttab; a.rows_per_target = L.P * DS;
    a.ngroups = ngroups; a.nsteps = nsteps;
    a.wtab = ta.wtab; a.ltab = ta.ltab; a.order = oa.order; a.dtab = ta.dtab;
    a.data = k.data; a.wscalar = k.wscalar; a.out = k.out;
    if (k.mode == GF_RESID_SCALAR) {
        BA_TRY(ctx->get_scratch(SL_PARTIAL, (size_t)k.C * L.T * a.ntile * sizeof(double), &p));
        a.partial = (double *)p;
    }
    int64_t nblocks = ngroups * L.T * a.ntile;
    const int order_knob = env_int("BEATAMD_GS_ORDER", 1);
    a.xcd_order = (ngroups > 1 && order_knob != 0) ? 1 : 0;
    if (a.xcd_order) nblocks = ((L.T * a.ntile + 7) / 8) * 8 * ngroups;
    BA_CHECK(nblocks < (int64_t)0x7fffffff, BEATAMD_EINVAL, "gfstack: batch too large");
    const int nth_knob = env_int("BEATAMD_GS_NTHINT", -1);
    const int nth = nth_knob >= 0 ? (nth_knob != 0) : (ngroups == 1);
    const size_t ring = std::max<size_t>((size_t)3 * nslot * 512, (size_t)GC_NCONS * 16 * GC_TPITCH);
    const size_t lds = GC_PARAM_BYTES + ring;
    BA_CHECK(lds <= 160 * 1024, BEATAMD_EINVAL, "internal: k_gfstack_ml row buffers exceed LDS");
    snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "%s<%d,%d>", runs ? "k_gfstack_runs" : "k_gfstack_ml", k.mode, nth);
    ctx->gs_ngtp = GT * L.P;
    ctx->gs_trep = L.T / Ttab;
    ctx->gs_N = L.N;
    ctx->gs_cg = GC_CG;
    {
        ScopedTimer tm(ctx, "gfstack");
        void (*kern)(GcArgs) = runs ? (nth ? k_gfstack_mlr<1, 0, 1> : k_gfstack_mlr<0, 0, 1>)
                                    : (nth ? k_gfstack_mlr<1, 0, 0> : k_gfstack_mlr<0, 0, 0>);
#if GM_NVARIANT > 1
        if (!runs) {
            const int var = env_int("BEATAMD_GM_VAR", 0);   // timing experiments (GM_ABLATIONS builds; wrong results)
            void (*vk[])(GcArgs) = {kern, k_gfstack_mlr<1, 1, 0>, k_gfstack_mlr<1, 2, 0>, k_gfstack_mlr<1, 3, 0>, k_gfstack_mlr<1, 4, 0>,
                                    k_gfstack_mlr<1, 5, 0>, k_gfstack_mlr<1, 6, 0>, k_gfstack_mlr<1, 7, 0>, k_gfstack_mlr<1, 8, 0>,
                                    k_gfstack_mlr<1, 9, 0>};
            if (var >= 1 && var < GM_NVARIANT) kern = vk[var];
        }
#endif
#if GR_NVARIANT > 1
        if (runs) {
            const int var = env_int("BEATAMD_GR_VAR", 0);
            void (*vk[])(GcArgs) = {kern, k_gfstack_mlr<1, 1, 1>, k_gfstack_mlr<1, 2, 1>, k_gfstack_mlr<1, 3, 1>, k_gfstack_mlr<1, 4, 1>, k_gfstack_mlr<1, 5, 1>,
                                    k_gfstack_mlr<1, 6, 1>, k_gfstack_mlr<1, 7, 1>, k_gfstack_mlr<1, 8, 1>, k_gfstack_mlr<1, 9, 1>, k_gfstack_mlr<1, 10, 1>};
            if (var >= 1 && var < GR_NVARIANT) kern = vk[var];
        }
#endif
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(1024), lds, ctx->stream, a);
    }
    BA_HIP(hipGetLastError());
    if (k.mode == GF_RESID_SCALAR) BA_TRY(launch_sum_tiles(ctx, a.partial, k.C * L.T, a.ntile, k.quad));
    return BEATAMD_OK;
}

}  // namespace beatamd
