// gfcell.hip -- multilinear Green's-function stacking for gfx950 with the rows of a cell in registers: k_gfstack_runs.
//
// Same arithmetic as k_gfstack (gfstack.hip; reference beat/ffi/base.py:607-709, multilinear
// branch :663-704): per (chain, target, sample) acc = fma(G[row_k], w_k, acc) over the four corner
// rows k of the chain's (duration, start-time) cell, patches ascending -- bitwise equal.
//
// With four rows per chain the lane <-> chain kernels of gfshared.hip read four LDS operands per
// FMA group and are bound by the LDS gather (19.4 ms per 512-chain launch on config 3).  Here the
// mapping is turned around:
//   workgroup  = (518-chain group, target, 64-sample tile) = 14 consumer + 2 loader wavefronts
//   consumer   = 37 chains; lane <-> sample; accumulator of chain j = VGPR pair ACC + 2j, selected through the
//                VGPR index register (M0, one scalar instruction per chain)
//   per step   : the chains of a wavefront are visited in CELL ORDER; the four rows of a cell are read from LDS
//                (contiguous 512-byte reads, no bank conflicts) only when the next chain opens a new cell and
//                applied by  v_fmac_f64_dpp acc[M0], w, x_k row_newbcast:(4q+k)
//   rows       : every distinct row segment of a step is fetched from HBM once by LDS-DMA
//                (global_load_lds_dwordx4, loader wavefronts) into a ring of three LDS buffers, two steps ahead.
// A STEP is one (patch, ROW PASS, slip variable) (round 5).  A row buffer holds GR_CAP = 104 row segments; the rows
// of a patch are laid out compactly in (duration line, start-time node) order, and a patch whose chain group
// touches more rows than that -- a library on a fine (duration x start-time) grid: 229 rows per patch for 512 prior
// chains on the reference tutorial's 17 x 41 grid -- is cut into passes along the duration axis: a chain takes part
// in the pass that holds its cell, in the other passes of the patch its position in the wavefront's walk is a PAD
// (zero weights into a scratch accumulator).  What a wavefront does is table driven (k_gm_tables): per step the
// loaders' request lines, the consumers' descriptor lines (scalar loads) and weight records.
// tools/gen_gfruns_asm.py generates the two wavefront programs (gfruns_asm.inc: the accumulators must be a
// contiguous physical register range, so they are register-allocated by hand); tools/gfcell_emu.py interprets
// them on the CPU (tests/test_gfcell_program.py).  Rounds 3 / 4 shipped two more consumer programs in this file
// (k_gfstack_cell, k_gfstack_ml); both were retired in round 5 (docs/design_history.md 3.1d-e keeps what they measured).
#include "kernels.hpp"
#include "gfruns_asm.inc"

namespace beatamd {

constexpr int GC_CG = GC_NCONS * GC_NCHAIN;    // chain slots per group (518)
constexpr int GC_WAVES = GC_NCONS + GC_NLOAD;
constexpr int GC_TB = 576;                      // threads of the table kernels (>= GC_CG)
constexpr int GC_TPITCH = 65 * 8;               // transposed misfit tile: row pitch in bytes
constexpr int GC_PARAM_BYTES = GC_WAVES * 128;
constexpr uint32_t GC_DEAD = 0xffffffffu;
// row slots of an LDS buffer: three buffers behind the wavefronts' parameter blocks in the CU's 160 KB
constexpr int GR_CAP = ((160 * 1024 - GC_PARAM_BYTES) / (3 * 512)) / 2 * 2;
// passes per patch the tables are sized for; a batch that needs more vsteps than P * GR_PASS_ALLOC for some
// (group, target) is stacked by k_gfstack instead (device-side flag, no host synchronisation)
constexpr int GR_PASS_ALLOC = 6;
constexpr int64_t GR_DENSE_MAX = 16384;         // D * (S + 1) the table kernel's LDS maps are sized for

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
    // batches of several groups: the chains of the whole batch cut into groups of `cg` chain slots by recursive bisection
    // along the key of the wider extent (k_gc_cut); group g takes members[g*cg ..] -- a compact piece of the
    // fault per group: fewer distinct rows to stage, fewer cells per wavefront.  nullptr: group g = chains g*cg .. as they come
    uint32_t *members;
    double *keyv[2];          // [C] both keys (scratch of the cut)
    int64_t cg, ngroups;
    int strips;               // A/B (BEATAMD_GC_GLOBAL=2): every level along key 0 = strips in the order of the first key
    int nbands;               // A/B (BEATAMD_GC_BANDS): bands of k_gc_order, 0 = from the group's extents
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
    a.keyv[0][c] = f0;
    a.keyv[1][c] = f1;
}

// The cut of a batch into its chain groups: ONE workgroup, the batch in LDS (C <= GC_MEMBERS_MAX chains, <= GC_GROUPS_MAX
// groups).  A range of groups [lo, hi) with more than one group is split into its first (hi - lo) / 2 groups and the rest
// along the key in which ITS chains spread wider (key 0 on a tie); the first part takes the (hi - lo) / 2 * cg chains that
// come first by (key, chain id).  Only the last group of the batch can be partial, and it stays the last: the ranges to
// the left are always full.  With the hypocentre as keys: 2048 chains -> four quadrants of the fault instead of four strips
// (tests/order_experiment.py: 16.9 instead of 19.2 distinct cells per group and patch, 6.2 instead of 5.2 chains per row
// read of a wavefront).  Per level: the ranges' extents by LDS atomics on order-preserving integer images of the keys, one
// bitonic sort of the whole batch by (range, key along the range's axis, chain id) -- ranges are position intervals, so
// the sort only moves chains inside their range -- and the split by position.  A last sort by (group, chain id).
// (The first version ranked every chain against every other, C * C comparisons per level from global memory: 0.7 ms at
// 2048 chains, 2 ms at 4096 -- 3 % of the step it schedules; this is ~20 us.)
constexpr int64_t GC_MEMBERS_MAX = 8192;
constexpr int GC_GROUPS_MAX = 64;
constexpr int GC_CUT_TB = 1024;

__device__ __forceinline__ unsigned long long gc_ordered(double f)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(f);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double gc_unordered(unsigned long long u)
{
    return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u));
}

// elements (key, tag = range lo << 16 | chain id) in ascending (range, key, id) order
__device__ __forceinline__ void gc_bitonic(double *skey, uint32_t *stag, int n, int tid)
{
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < n / 2; q += GC_CUT_TB) {
                const int i = 2 * j * (q / j) + (q % j), l = i + j;
                const bool up = (i & k) == 0;
                const double ki = skey[i], kl = skey[l];
                const uint32_t ti = stag[i], tl = stag[l];
                const bool l_first = ((tl ^ ti) >> 16) ? tl < ti : (kl != ki ? kl < ki : tl < ti);
                if (l_first == up) {
                    skey[i] = kl; skey[l] = ki;
                    stag[i] = tl; stag[l] = ti;
                }
            }
            __syncthreads();
        }
}

__global__ void __launch_bounds__(GC_CUT_TB) k_gc_cut(GcOrderArgs a, int n, int64_t chunk)
{
    extern __shared__ __attribute__((aligned(16))) double skey[];          // [n]   n = C rounded up to a power of two
    uint32_t *stag = reinterpret_cast<uint32_t *>(skey + n);               // [n]
    uint16_t *slo = reinterpret_cast<uint16_t *>(stag + n), *shi = slo + n;   // [n] the group range a POSITION belongs to
    __shared__ unsigned long long ext[4][GC_GROUPS_MAX];                   // of the range that starts at group lo
    __shared__ int axis[GC_GROUPS_MAX];
    const int tid = threadIdx.x;
    // batches beyond GC_MEMBERS_MAX chains / GC_GROUPS_MAX groups: one workgroup per CHUNK of whole groups as the chains
    // come (round 6: BEAT's recommended n_chains reaches 10 000); chain ids below are relative to the chunk
    const int64_t base = (int64_t)blockIdx.x * chunk;
    const int C = (int)min(chunk, a.C - base), cg = (int)a.cg;
    const int ng = (C + cg - 1) / cg;
    a.keyv[0] += base; a.keyv[1] += base; a.members += base;
    for (int i = tid; i < n; i += GC_CUT_TB) {
        stag[i] = i < C ? (uint32_t)i : 0xffffffffu;                        // (pads: behind every range)
        slo[i] = i < C ? 0 : 0xffff;
        shi[i] = i < C ? (uint16_t)ng : 0xffff;
    }
    __syncthreads();
    for (int widest = ng; widest > 1; widest -= widest / 2) {              // (the wider part of a split)
        for (int g = tid; g < ng; g += GC_CUT_TB) {
            ext[0][g] = ~0ull; ext[1][g] = 0ull; ext[2][g] = ~0ull; ext[3][g] = 0ull;
        }
        __syncthreads();
        for (int i = tid; i < C; i += GC_CUT_TB) {
            const int lo = slo[i];
            if (shi[i] - lo <= 1) continue;
            const uint32_t c = stag[i] & 0xffffu;
            const unsigned long long u0 = gc_ordered(a.keyv[0][c]), u1 = gc_ordered(a.keyv[1][c]);
            atomicMin(&ext[0][lo], u0); atomicMax(&ext[1][lo], u0);
            atomicMin(&ext[2][lo], u1); atomicMax(&ext[3][lo], u1);
        }
        __syncthreads();
        for (int g = tid; g < ng; g += GC_CUT_TB)
            axis[g] = (!a.strips && ext[1][g] >= ext[0][g] &&
                       gc_unordered(ext[3][g]) - gc_unordered(ext[2][g]) > gc_unordered(ext[1][g]) - gc_unordered(ext[0][g])) ? 1 : 0;
        __syncthreads();
        for (int i = tid; i < n; i += GC_CUT_TB) {
            double f = 0.0;
            if (i < C) {
                const int lo = slo[i];
                const uint32_t c = stag[i] & 0xffffu;
                if (shi[i] - lo > 1) f = a.keyv[axis[lo]][c];
                stag[i] = (uint32_t)lo << 16 | c;
            }
            skey[i] = f;
        }
        __syncthreads();
        gc_bitonic(skey, stag, n, tid);
        for (int i = tid; i < C; i += GC_CUT_TB) {
            const int lo = slo[i], hi = shi[i];
            if (hi - lo <= 1) continue;
            const int half = (hi - lo) / 2;
            if (i < (lo + half) * cg) shi[i] = (uint16_t)(lo + half);
            else slo[i] = (uint16_t)(lo + half);
        }
        __syncthreads();
    }
    // inside a group: by chain id
    for (int i = tid; i < n; i += GC_CUT_TB) {
        if (i < C) stag[i] = (uint32_t)slo[i] << 16 | (stag[i] & 0xffffu);
        skey[i] = 0.0;
    }
    __syncthreads();
    gc_bitonic(skey, stag, n, tid);
    for (int i = tid; i < C; i += GC_CUT_TB) a.members[i] = (uint32_t)base + (stag[i] & 0xffffu);
}

// (round 6: two bitonic sorts of the group's 518 slots in LDS -- by the first key, then by (band, second key) -- instead of
// two O(518^2) ranking loops per thread: the single workgroup of a 512-chain batch took 103 us, a twentieth of a step on
// 120-sample traces.  Ties go by slot as before: the order is the same.)
__global__ void __launch_bounds__(GC_CUT_TB) k_gc_order(GcOrderArgs a)
{
    constexpr int NS = 1024;                       // slots sorted: GC_CG rounded up to a power of two
    static_assert(GC_CG <= NS && NS <= 2 * GC_CUT_TB, "k_gc_order: one workgroup sorts a group");
    __shared__ double skey[NS];
    __shared__ uint32_t stag[NS];
    __shared__ double kb[GC_CG];
    __shared__ uint32_t cids[GC_CG];
    __shared__ double ext[4][GC_CUT_TB / 64];
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
    if (slot) { kb[tid] = f1; cids[tid] = live ? (uint32_t)c : GC_DEAD; }
    // dead slots and the pads of the sort: behind every live slot (range 0xffff)
    for (int i = tid; i < NS; i += GC_CUT_TB) {
        skey[i] = (i == tid && live) ? f0 : 0.0;
        stag[i] = (i == tid && live) ? (uint32_t)tid : (0xffff0000u | (uint32_t)(i & 0xffff));
    }
    const int nlive = (int)min((int64_t)GC_CG, a.C - (int64_t)blockIdx.x * GC_CG);
    // extents of the two keys over the live slots: wavefront reductions, then the partial results
    {
        const double s0 = __shfl(f0, 0, 64), s1 = __shfl(f1, 0, 64);      // (a live lane's keys for the dead ones: slot 0 of
        double m0 = live ? f0 : s0, M0 = m0, m1 = live ? f1 : s1, M1 = m1;  //  wavefront 0 is live; later wavefronts see below)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m0 = fmin(m0, __shfl_xor(m0, off, 64)); M0 = fmax(M0, __shfl_xor(M0, off, 64));
            m1 = fmin(m1, __shfl_xor(m1, off, 64)); M1 = fmax(M1, __shfl_xor(M1, off, 64));
        }
        if ((tid & 63) == 0) { ext[0][tid >> 6] = m0; ext[1][tid >> 6] = M0; ext[2][tid >> 6] = m1; ext[3][tid >> 6] = M1; }
    }
    __syncthreads();
    gc_bitonic(skey, stag, NS, tid);               // slots in (first key, slot) order: position = rank
    const int nwl = (nlive + 63) / 64;             // wavefronts that hold a live slot (their lane 0 is live)
    double mn0 = ext[0][0], mx0 = ext[1][0], mn1 = ext[2][0], mx1 = ext[3][0];
    for (int q = 1; q < nwl; q++) {
        mn0 = fmin(mn0, ext[0][q]); mx0 = fmax(mx0, ext[1][q]);
        mn1 = fmin(mn1, ext[2][q]); mx1 = fmax(mx1, ext[3][q]);
    }
    const int nw = (nlive + GC_NCHAIN - 1) / GC_NCHAIN;
    // bands of whole wavefronts by the first key, as many as make a wavefront's chains a SQUARE piece of the group's
    // extent in the two keys: nb / (nw / nb) = e0 / e1 (a group that covers the fault: 4 bands of 14 wavefronts; a strip
    // four times as long in the second key: 2).  Without caller keys (start-time indices): five, as measured in round 4
    int nb = 5;
    if (a.key[0].base && a.key[1].base) {
        const double e0 = mx0 - mn0, e1 = mx1 - mn1;
        double r = e1 > 0.0 ? (double)nw * e0 / e1 : (double)nw * (double)nw;
        if (!(r <= (double)nw * (double)nw)) r = (double)nw * (double)nw;      // (extents that overflow, inf / inf)
        nb = max(1, min(nw, (int)rint(sqrt(r))));
        if (a.nbands > 0) nb = min(nw, a.nbands);
    }
    // second sort: (band of the slot's rank, second key, slot)
    uint32_t t2[NS / GC_CUT_TB];
    double k2[NS / GC_CUT_TB];
    for (int i = tid, u = 0; i < NS; i += GC_CUT_TB, u++) {
        const uint32_t tg = stag[i];
        const bool lv = (tg >> 16) == 0;           // (live slots sorted to positions 0 .. nlive-1)
        const uint32_t sl = tg & 0xffffu;
        t2[u] = lv ? ((uint32_t)((i / GC_NCHAIN) * nb / nw) << 16 | sl) : tg;
        k2[u] = lv ? kb[sl] : 0.0;
    }
    __syncthreads();
    for (int i = tid, u = 0; i < NS; i += GC_CUT_TB, u++) { stag[i] = t2[u]; skey[i] = k2[u]; }
    __syncthreads();
    gc_bitonic(skey, stag, NS, tid);
    if (slot) {
        const uint32_t tg = stag[tid];
        a.order[(int64_t)blockIdx.x * GC_CG + tid] = (tg >> 16) == 0xffffu ? GC_DEAD : cids[tg & 0xffffu];
    }
}

__global__ void __launch_bounds__(256) k_members_pad(uint32_t *members, int64_t C, int64_t padded)
{
    const int64_t i = C + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < padded) members[i] = GC_DEAD;
}

// the cut of a batch into ngroups groups of cg chain slots -> oa.members (scratch at `p`: members [padded], two key vectors)
static size_t gc_cut_bytes(int64_t C, int64_t padded) { return (size_t)padded * 4 + (size_t)C * 16 + 64; }
// chains per chunk of the cut: whole groups, at most GC_GROUPS_MAX of them and GC_MEMBERS_MAX chains (8192 for 512-chain
// groups, 7770 = 15 x 518 for the runs kernel's)
static int64_t gc_cut_chunk(int64_t cg) { return std::max<int64_t>(1, std::min<int64_t>(GC_MEMBERS_MAX / cg, GC_GROUPS_MAX)) * cg; }
static bool gc_cut_applicable(int64_t C, int64_t ngroups) { return ngroups > 0 && C > 0 && C / ngroups <= GC_MEMBERS_MAX; }

static int launch_gc_cut(beatamd_ctx *ctx, GcOrderArgs &oa, void *p, int64_t cg, int64_t ngroups, int64_t padded)
{
    oa.members = (uint32_t *)p;
    oa.keyv[0] = reinterpret_cast<double *>(((uintptr_t)(oa.members + padded) + 7) & ~(uintptr_t)7);
    oa.keyv[1] = oa.keyv[0] + oa.C;
    oa.cg = cg; oa.ngroups = ngroups;
    hipLaunchKernelGGL(k_gc_key0, dim3((unsigned)((oa.C + 255) / 256)), dim3(256), 0, ctx->stream, oa);
    const int64_t chunk = gc_cut_chunk(cg);
    int n = 2;
    while (n < std::min<int64_t>(oa.C, chunk)) n <<= 1;
    const size_t lds = (size_t)n * (8 + 4 + 2 + 2);
    BA_HIP(hipFuncSetAttribute((const void *)k_gc_cut, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_gc_cut, dim3((unsigned)((oa.C + chunk - 1) / chunk)), dim3(GC_CUT_TB), lds, ctx->stream, oa, n, chunk);
    if (padded > oa.C)
        hipLaunchKernelGGL(k_members_pad, dim3((unsigned)((padded - oa.C + 255) / 256)), dim3(256), 0, ctx->stream, oa.members, oa.C, padded);
    return BEATAMD_OK;
}

int launch_chain_members(beatamd_ctx *ctx, int64_t C, const ChainVec key[2], int64_t cg, int64_t ngroups, const uint32_t **members,
                         int strips)
{
    *members = nullptr;
    if (!gc_cut_applicable(C, ngroups) || !key[0].base || !key[1].base) return BEATAMD_OK;
    void *p = nullptr;
    const int64_t padded = ngroups * cg;
    BA_TRY(ctx->get_scratch(SL_GS_ORDER, gc_cut_bytes(C, padded), &p));
    GcOrderArgs oa;
    memset(&oa, 0, sizeof(oa));
    oa.C = C; oa.T = 1; oa.P = 1; oa.S = 1;
    oa.key[0] = key[0]; oa.key[1] = key[1];
    oa.strips = strips;
    BA_TRY(launch_gc_cut(ctx, oa, p, cg, ngroups, padded));
    BA_HIP(hipGetLastError());
    *members = oa.members;
    return BEATAMD_OK;
}

// launches the chain order of a batch into oa.order (scratch for members / keys behind it)
static int launch_gc_order(beatamd_ctx *ctx, GcOrderArgs &oa, int64_t ngroups, const GfKnobs &kn)
{
    void *p = nullptr;
    const size_t norder = (size_t)(ngroups * GC_CG + 64);
    const bool global = oa.sort && ngroups > 1 && gc_cut_applicable(oa.C, ngroups) && GfKnobs::get(kn.gc_global, 1) != 0;
    BA_TRY(ctx->get_scratch(SL_GC_ORDER, norder * sizeof(uint32_t) + (global ? gc_cut_bytes(oa.C, oa.C) + 64 : 0), &p));
    oa.order = (uint32_t *)p;
    oa.members = nullptr;
    oa.strips = GfKnobs::get(kn.gc_global, 1) == 2;
    oa.nbands = GfKnobs::get(kn.gc_bands, 0);
    if (global) {
        void *q = reinterpret_cast<void *>(((uintptr_t)(oa.order + norder) + 7) & ~(uintptr_t)7);
        BA_TRY(launch_gc_cut(ctx, oa, q, GC_CG, ngroups, oa.C));
    }
    hipLaunchKernelGGL(k_gc_order, dim3((unsigned)ngroups), dim3(GC_CUT_TB), 0, ctx->stream, oa);
    return BEATAMD_OK;
}

// ---------------------------------------------------------------------------- tables
// DENSE SLOT of a library row of a patch: slot(d, s') = d*(S+1) + s', s' = s + 1; s' = 0 of a duration line stands for a
// copy of its LAST start-time node -- the floor node of ceil node 0 (python negative index, base.py:513-517: ceil - 1 =
// -1 -> S-1).  A chain with ceil nodes (dc, sc) and floor line df uses dense slots A = df*(S+1) + sc, A + 1 (floor
// line: floor / ceil start time) and B = dc*(S+1) + sc, B + 1.  A pass stages its dense slots COMPACTLY in ascending
// order: consecutive dense slots stay consecutive, so a chain's rows are still two LDS addresses.
struct GmTabArgs {
    int64_t C, T, P, D, S, DS;    // T: targets the tables are built for (1 or all)
    int nvar, cap;                // slip variables: step = vstep * nvar + variable (same rows, the variable's slips)
    int64_t vmax, smax;           // vsteps / steps per (group, target) the tables are strided by
    const uint32_t *rowoff;       // [C,T,P,4] global row ids (k_gf_tables: cc, fc, cf, ff)
    const double *fac;            // [C,T,P,4]
    ChainVec slips[4];
    const uint32_t *order;        // [ngroups*GC_CG]
    uint32_t *npass;              // [gtp] passes of the patch                                (count phase out)
    uint8_t *cpass;               // [gtp][GC_CG] pass of every chain slot (0xff: no chain)  (count phase out, fill in)
    const uint32_t *voff;         // [gtp] first vstep of the patch; nullptr: one pass per patch, vstep = patch
    const int *ovf;               // fill: nonzero = some (group, target) needs more vsteps than vmax: nothing is written
    char *wtab;                   // [(g*T+t)][consumer][step 0..smax][GR_WSTRIDE] record pairs of 2 x 16 weights
    uint32_t *ltab;               // [(g*T+t)][step 0..smax+2][loader][GC_LTABDW]: count, first row of the patch, requests
    uint32_t *dtab;               // [(g*T+t)][consumer][step 0..smax][GR_DLINE] position descriptors (scalar loads)
    uint32_t *ucount;             // [gtp] row segments the loaders move for the patch, all passes (statistics)
    int64_t R;                    // patch split: slot t covers patches (t % R) * P + p of the real model (slips)
    int64_t ngtp;                 // (group, target, patch) items
};

__global__ void __launch_bounds__(256) k_gm_scan(const uint32_t *npass, uint32_t *voff, uint32_t *nv, int64_t P, int64_t vmax,
                                                 int *ovf)
{
    __shared__ uint32_t part[256];
    const int tid = threadIdx.x;
    const int64_t gt = blockIdx.x;
    uint32_t run = 0;
    for (int64_t base = 0; base < P; base += 256) {
        const int64_t p = base + tid;
        const uint32_t x = p < P ? npass[gt * P + p] : 0u;
        part[tid] = x;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int k = 0; k < 256; k++) {
            const uint32_t y = part[k];
            if (k < tid) before += y;
            total += y;
        }
        if (p < P) voff[gt * P + p] = run + before;
        run += total;
        __syncthreads();
    }
    if (tid == 0) {
        nv[gt] = run;
        if ((int64_t)run > vmax) atomicOr(ovf, 1);
    }
}

// number of row requests a pass of n slots over nlines duration lines can take at most: pairs of ascending rows less than
// 256 apart, singles at a line's wrap slot, at jumps and at the end
__device__ __forceinline__ int gm_req_bound(int n, int nlines, int64_t S) { return S > 255 ? n : n / 2 + 2 * nlines + 1; }

// The row passes of a patch from the bitsets of its chain group (one thread): C / F = slots used on a duration line as ceil
// / floor line, X = cells by ceil line.  Passes along the duration axis; a line whose cells alone exceed a buffer is cut along
// the start-time axis (linesplit, mark = pass of cell (d, s)).  -> number of passes; linepass[d] = pass of the cells with
// ceil line d.  (numpy twin: tools/gfcell_emu.py gm_passes)
__device__ int gm_count_passes(const uint32_t *bC, const uint32_t *bF, const uint32_t *bX, uint8_t *mark, uint8_t *linepass,
                               uint8_t *linesplit, const int D, const int W, const uint32_t S, const int64_t S1, const int cap,
                               const int64_t S64)
{
    auto pop = [&](const uint32_t *x, const uint32_t *y) {   // |x u y| (y nullable)
        int n = 0;
        for (int k = 0; k < W; k++) n += __popc(x[k] | (y ? y[k] : 0u));
        return n;
    };
    auto any = [&](const uint32_t *x) {
        for (int k = 0; k < W; k++)
            if (x[k]) return true;
        return false;
    };
    auto fl = [&](int d) { return d == 0 ? D - 1 : d - 1; };   // floor line of ceil line d (base.py:513-517)
    int cur = -1, n = 0, nlines = 0, first = -1;
    bool open = false;
    for (int d = 0; d < D; d++) {
        if (!any(bX + d * W)) continue;
        const int f = fl(d);
        // slots the cells of ceil line d add: their ceil line (with what it already holds as the floor line of
        // line d + 1 -- only the wrap: line D-1 under ceil line 0) and their floor line (with what it holds as
        // a ceil line of the pass)
        const bool wrap_c = open && d == D - 1 && first == 0 && D > 1;      // line D-1 already holds F[D-1]
        const bool floor_in = open && f != d && ((f >= first && f < d));   // line f is a ceil line of the pass
        int add, lines_add;
        if (f == d) {   // one duration node: both usages on one line
            add = pop(bC + d * W, bF + d * W);
            lines_add = 1;
        } else {
            add = pop(bC + d * W, wrap_c ? bF + d * W : nullptr) - (wrap_c ? pop(bF + d * W, nullptr) : 0);
            add += floor_in ? pop(bF + f * W, bC + f * W) - pop(bC + f * W, nullptr) : pop(bF + f * W, nullptr);
            lines_add = (wrap_c ? 0 : 1) + ((floor_in && any(bC + f * W)) ? 0 : 1);
        }
        const int alone = (f == d) ? add : pop(bC + d * W, nullptr) + pop(bF + f * W, nullptr);
        if (open && (n + add > cap || gm_req_bound(n + add, nlines + lines_add, S64) > GC_NLOAD * GC_LREQ)) open = false;
        if (!open) {
            if (alone > cap || gm_req_bound(alone, f == d ? 1 : 2, S64) > GC_NLOAD * GC_LREQ) {
                // the cells of this line alone do not fit: cut the line along the start-time axis; every cell
                // {sc, sc + 1} on both lines: 2 slots per distinct node
                linesplit[d] = 1;
                int m = 0;
                cur++;
                int prev = -2;
                for (uint32_t s_ = 0; s_ < S; s_++) {
                    if (!((bX[d * W + (s_ >> 5)] >> (s_ & 31)) & 1u)) continue;
                    const int addc = ((int)s_ == prev + 1 ? 1 : 2) * (f == d ? 1 : 2);
                    if (m && (m + addc > cap || gm_req_bound(m + addc, 2, S64) > GC_NLOAD * GC_LREQ)) {
                        cur++;
                        m = 0;
                        prev = -2;
                    }
                    m += ((int)s_ == prev + 1 ? 1 : 2) * (f == d ? 1 : 2);
                    prev = (int)s_;
                    mark[d * S1 + s_] = (uint8_t)cur;     // (mark: pass of cell (d, s) of a split line)
                }
                continue;
            }
            cur++;
            open = true;
            first = d;
            n = alone;
            nlines = f == d ? 1 : 2;
        } else {
            n += add;
            nlines += lines_add;
        }
        linepass[d] = (uint8_t)cur;
    }
    return cur + 1;
}

// one workgroup per (group, target, patch); thread <-> chain slot of the group order.
// FILL = 0: the passes of the patch (npass, cpass);  FILL = 1: the tables of its steps
template <int FILL>
__global__ void __launch_bounds__(GC_TB) k_gm_tables(GmTabArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    // dynamic: bitsets over s' per duration line (W words each) -- C: used as ceil line, F: used as floor line, X: cells
    // (bit sc of line dc) -- then mark[dense] / cidx[dense] bytes of the pass in hand
    const int W = (int)((a.S + 1 + 31) / 32);
    const int64_t S1 = a.S + 1, dense_n = a.D * S1;
    uint32_t *bC = dyn, *bF = bC + a.D * W, *bX = bF + a.D * W;
    uint8_t *mark = reinterpret_cast<uint8_t *>(bX + a.D * W);
    uint8_t *cidx = mark + ((dense_n + 3) & ~(int64_t)3);
    uint8_t *linepass = cidx + ((dense_n + 3) & ~(int64_t)3);          // [D] pass of the cells with ceil line d (unsplit lines)
    uint8_t *linesplit = linepass + ((a.D + 3) & ~(int64_t)3);          // [D] the line is cut along the start-time axis
    __shared__ uint32_t keys[GC_CG];          // (B << 16) | A of the chain slot, ~0: no chain
    __shared__ uint8_t cps[GC_CG];            // pass of the chain slot (0xff: none)
    __shared__ uint16_t clist[128];           // compact slot -> dense slot of the pass in hand
    __shared__ uint32_t poskey[GC_CG];        // per wavefront: key at walk position r of the pass in hand
    __shared__ uint32_t wsum[GC_TB / 64 + 1];
    __shared__ uint32_t rsum[2][GC_TB / 64];
    __shared__ int sh_npass, sh_n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t gtp = xcd_items8(blockIdx.x, gridDim.x);
    const int64_t p = gtp % a.P;
    const int64_t gt = gtp / a.P;
    const int64_t t = gt % a.T;
    const int64_t g = gt / a.T;
    if constexpr (FILL) {
        if (a.ovf && *a.ovf) return;
    }
    const bool slot = tid < GC_CG;
    const uint32_t cid = slot ? a.order[g * GC_CG + tid] : GC_DEAD;
    const bool live = cid != GC_DEAD;
    const int64_t c = live ? (int64_t)cid : 0;
    const int64_t row0 = (t * a.P + p) * a.DS;
    const uint32_t S = (uint32_t)a.S;
    // (blocks of several consecutive patches per workgroup with the next patch's entries prefetched were measured in round
    // 6: 8 patches 0.58 ms against 0.42 for configs[3] -- the per-patch latency is the cost, not the launch; k_gm_tables_w
    // below is the answer to it)

    uint32_t dc = 0, sc = 0, df = 0, sa = 0, sb = 0;
    double fr[4] = {0, 0, 0, 0};
    if (live) {
        const int64_t e = ((c * a.T + t) * a.P + p) * 4;
        const uint32_t v0 = a.rowoff[e] - (uint32_t)row0;        // (ceil d, ceil s)
        const uint32_t v2 = a.rowoff[e + 2] - (uint32_t)row0;    // (floor d, ceil s)
        dc = v0 / S; sc = v0 % S; df = v2 / S;
        sb = dc * (uint32_t)S1 + sc;     // the floor node of ceil node sc is slot sc of the line (sc = 0: the wrap copy)
        sa = df * (uint32_t)S1 + sc;
        if constexpr (FILL)
            for (int k = 0; k < 4; k++) fr[k] = a.fac[e + k];
    }
    const uint32_t key = live ? ((sb << 16) | sa) : 0xffffffffu;
    if (slot) keys[tid] = key;

    bool have_passes = false;
    if constexpr (FILL) have_passes = a.voff != nullptr;
    if (FILL && have_passes) {
        if (slot) cps[tid] = a.cpass[gtp * GC_CG + tid];
        if (tid == 0) sh_npass = (int)a.npass[gtp];
        __syncthreads();
    } else if (FILL) {
        if (slot) cps[tid] = live ? 0 : 0xff;
        if (tid == 0) sh_npass = 1;
        __syncthreads();
    } else {
        // ---------------- count phase: passes along the duration axis
        for (int i = tid; i < 3 * a.D * W; i += GC_TB) dyn[i] = 0;
        for (int i = tid; i < a.D; i += GC_TB) { linepass[i] = 0; linesplit[i] = 0; }
        __syncthreads();
        if (live) {
            atomicOr(&bC[dc * W + (sc >> 5)], 1u << (sc & 31));
            atomicOr(&bC[dc * W + ((sc + 1) >> 5)], 1u << ((sc + 1) & 31));
            atomicOr(&bF[df * W + (sc >> 5)], 1u << (sc & 31));
            atomicOr(&bF[df * W + ((sc + 1) >> 5)], 1u << ((sc + 1) & 31));
            atomicOr(&bX[dc * W + (sc >> 5)], 1u << (sc & 31));
        }
        __syncthreads();
        if (tid == 0) sh_npass = gm_count_passes(bC, bF, bX, mark, linepass, linesplit, (int)a.D, W, S, S1, a.cap, a.S);
        __syncthreads();
        if (slot) {
            uint8_t cp = 0xff;
            if (live) cp = linesplit[dc] ? mark[dc * S1 + sc] : linepass[dc];
            a.cpass[gtp * GC_CG + tid] = cp;
        }
        // (pass ids are bytes: a patch cut into more than 250 passes -- buffers of a few slots in tests -- counts as an
        // overflow of the tables: the streaming kernel stands in)
        if (tid == 0) a.npass[gtp] = sh_npass > 250 ? 0x100000u : (uint32_t)sh_npass;
        return;
    }
    if constexpr (FILL) {
    // ---------------- fill phase
    const int npass = sh_npass;
    const int64_t v0 = a.voff ? (int64_t)a.voff[gtp] : p;
    const int w = tid / GC_NCHAIN, j = tid % GC_NCHAIN;
    const uint8_t mypass = slot ? cps[tid] : 0xff;
    uint32_t moved = 0;
    for (int k = 0; k < npass; k++) {
        // dense slots of the pass -> compact slots
        for (int64_t i = tid; i < ((dense_n + 3) >> 2); i += GC_TB) reinterpret_cast<uint32_t *>(mark)[i] = 0;
        __syncthreads();
        const bool mine = slot && live && mypass == (uint8_t)k;
        if (mine) { mark[sa] = 1; mark[sa + 1] = 1; mark[sb] = 1; mark[sb + 1] = 1; }   // benign race: every writer stores 1
        __syncthreads();
        uint32_t run = 0;
        for (int64_t base = 0; base < dense_n; base += GC_TB) {
            const int64_t i = base + tid;
            const uint32_t f = (i < dense_n) ? mark[i] : 0u;
            const uint64_t m = __ballot(f != 0);
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int q = 0; q < GC_TB / 64; q++) {
                const uint32_t x = wsum[q];
                if (q < wv) before += x;
                total += x;
            }
            if (f) {
                const uint32_t ps = run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                cidx[i] = (uint8_t)ps;
                if (ps < 128) clist[ps] = (uint16_t)i;
            }
            run += total;
            __syncthreads();
        }
        const int n = (int)run;       // <= cap (count phase)
        moved += run;
        // ---- row requests: pairs of compact neighbours whose rows ascend by less than 256, singles otherwise; elements
        // 0..n-1 on the first two wavefronts
        auto row_of = [&](uint32_t sl) { const uint32_t d = sl / (uint32_t)S1, s1 = sl % (uint32_t)S1; return d * S + (s1 ? s1 - 1 : S - 1); };
        uint32_t req = 0;
        int rq_idx = -1;
        {
            const int i = tid;
            const bool in = i < n && i < 128;
            const uint32_t r_i = in ? row_of(clist[i]) : 0u;
            const uint32_t r_p = (in && i > 0) ? row_of(clist[i - 1]) : 0u;
            const bool pairable = in && i > 0 && r_i > r_p && r_i - r_p < 256;   // may follow its predecessor in a pair
            // start of the run of pairable elements the element is in: inclusive max scan of (run start ? i : 0)
            int st = (in && !pairable) ? i : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(st, off, 64);
                if (lane >= off) st = max(st, o);
            }
            if (lane == 63) rsum[0][wv] = (uint32_t)st;
            __syncthreads();
            if (wv == 1) st = max(st, (int)rsum[0][0]);
            const bool leads = in && (((i - st) & 1) == 0);        // first row of a pair, or a single
            uint32_t r_n = 0;
            bool pair = false;
            if (leads && i + 1 < n) {
                r_n = row_of(clist[i + 1]);
                pair = r_n > r_i && r_n - r_i < 256;
            }
            const uint64_t m = __ballot(leads);
            if (lane == 0) rsum[1][wv] = (uint32_t)__popcll(m);
            __syncthreads();
            if (leads) {
                rq_idx = (int)((wv == 1 ? rsum[1][0] : 0u) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
                req = r_i | ((pair ? r_n - r_i : 0u) << 16) | ((uint32_t)i << 24);
                if (rq_idx >= GC_NLOAD * GC_LREQ) rq_idx = -1;     // (cannot happen: the count phase bounds the requests)
            }
            if (tid == 0) sh_n = min((int)(rsum[1][0] + rsum[1][1]), GC_NLOAD * GC_LREQ);
        }
        __syncthreads();
        const int nreq = sh_n;
        // ---- walk positions of the wavefronts: the chains of the pass in cell order, the others (pads) behind them
        int r = 0, nmem_w = 0;
        if (slot) {
            int nmem = 0, rk = 0, rp = 0;
            for (int q = 0; q < GC_NCHAIN; q++) {
                const int o = w * GC_NCHAIN + q;
                const bool om = cps[o] == (uint8_t)k;
                const uint32_t ok = keys[o];
                nmem += om;
                if (mine) rk += om && ((ok < key) || (ok == key && q < j));
                else rp += !om && q < j;
            }
            r = mine ? rk : nmem + rp;
            poskey[w * GC_NCHAIN + r] = mine ? key : 0xffffffffu;
            nmem_w = nmem;
        }
        __syncthreads();
        const bool next_opens = mine && r + 1 < GC_NCHAIN && poskey[w * GC_NCHAIN + r + 1] != 0xffffffffu &&
                                poskey[w * GC_NCHAIN + r + 1] != key;
        const uint32_t ca = mine ? cidx[sa] : 0u, cb = mine ? cidx[sb] : 0u;
        for (int iv = 0; iv < a.nvar; iv++) {
            const int64_t s = (v0 + k) * a.nvar + iv;
            if (rq_idx >= 0) {
                const int ll = rq_idx % GC_NLOAD;
                a.ltab[((gt * (a.smax + 3) + s) * GC_NLOAD + ll) * GC_LTABDW + 2 + rq_idx / GC_NLOAD] = req;
            }
            if (tid < GC_NLOAD) {
                uint32_t *h = a.ltab + ((gt * (a.smax + 3) + s) * GC_NLOAD + tid) * GC_LTABDW;
                h[0] = nreq > tid ? (uint32_t)((nreq - tid + GC_NLOAD - 1) / GC_NLOAD) : 0u;
                h[1] = (uint32_t)(p * a.DS);
            }
            if (!slot) continue;
            const uint32_t ring = (uint32_t)((s % 3) * a.cap);
            const double sl = mine ? a.slips[iv].base[c * a.slips[iv].stride + a.slips[iv].off + (t % a.R) * a.P + p] : 0.0;
            const int q = r & 3;
            // weights only: a 256-byte record pair serves eight positions, entry e = {weight e of record 2p, of record 2p + 1}
            char *rec = a.wtab + ((gt * GC_NCONS + w) * (a.smax + 1) + s) * (int64_t)GR_WSTRIDE + (r >> 3) * GR_PAIR + ((r >> 2) & 1) * 8;
            for (int kk = 0; kk < 4; kk++)
                *reinterpret_cast<double *>(rec + (4 * q + kk) * 16) = mine ? fr[kk] * sl : 0.0;     // base.py:676-679 x slip, as k_gfstack
            uint32_t *dl = a.dtab + ((gt * GC_NCONS + w) * (a.smax + 1) + s) * GR_DLINE + (r < GR_NHALF ? 2 * r : GR_DHALF + 2 * (r - GR_NHALF));
            // a pad (a chain of another pass, an empty chain slot): zero weights into the scratch accumulator, no row reads
            dl[0] = (uint32_t)GR_D_BASE | (uint32_t)(mine ? j : GC_SCRATCH) | ((next_opens ? 1u : 0u) << 31);
            dl[1] = (ring + ca) | ((ring + cb) << 16);
            // chains of the wavefront in this step: the walk leaves the step at the first checkpoint behind them
            if (j == 0) a.dtab[((gt * GC_NCONS + w) * (a.smax + 1) + s) * GR_DLINE + GR_D_NCH] = (uint32_t)nmem_w;
        }
        __syncthreads();
    }
    if (tid == 0) a.ucount[gtp] = moved;
    // the request lines behind the last step of the (group, target) stay empty: the loaders read three steps ahead
    if (p == a.P - 1 && tid < 3 * GC_NLOAD) {
        const int64_t s = (v0 + npass) * a.nvar + tid / GC_NLOAD;
        uint32_t *h = a.ltab + ((gt * (a.smax + 3) + s) * GC_NLOAD + tid % GC_NLOAD) * GC_LTABDW;
        h[0] = 0;
        h[1] = 0;
    }
    }   // FILL
}

// ---------------------------------------------------------------------------- tables, one WAVEFRONT per patch (round 6)
// The same tables, entry for entry, as k_gm_tables -- but a (group, target, patch) is the work of ONE wavefront that visits
// the 518 chain slots nine per lane, with its own piece of LDS and no workgroup barrier anywhere.  k_gm_tables is a chain of
// ~10 barriers and ~5 dependent round trips to memory per patch on nine wavefronts that mostly wait for each other: with
// two such workgroups per CU the 6 800 (slot, patch) pairs of configs[3] (17 station slots x 400 patches) took 0.43 ms -- a quarter of
// the step at 120 samples per trace, whatever the block of patches per workgroup (8 consecutive patches per workgroup, prefetched, were SLOWER:
// 0.58 ms: it is the per-patch latency, not the launch).  Here a CU holds 20 patches in flight and a wavefront never waits
// for another.  Libraries up to GW_DENSE_MAX dense slots per patch (one bitset word per lane); beyond that k_gm_tables.
constexpr int GW_NW = 4;                       // wavefronts (= patches) per workgroup
constexpr int64_t GW_DENSE_MAX = 2048;         // D * (S + 1)
constexpr int GW_KC = 520 * 4;                 // bytes of a [GC_CG] array of words
constexpr uint32_t GW_PAD = 0x3fffffu;         // 22-bit key of a slot that takes no part (no chain / chain of another pass)

__device__ __forceinline__ void wave_sync()
{
    // the lanes of a wavefront run in lockstep: wait for its LDS operations, keep the compiler from moving memory
    // operations across
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// LDS of a wavefront: kc[GC_CG] (pass of the chain slot << 22 | 22-bit key: (B << 11) | A, GW_PAD: no chain), then
//   count: the bitsets C / F / X [3 D W], mark [dense], linepass / linesplit [D]
//   fill : skp[GC_CG] sort words, nmw[32], clist[128] u16, bits[64], wpre[64], the image of a consumer's weight record
//          (1280 bytes) and descriptor line (320)                                     (6560 bytes: 24 wavefronts per CU)
static size_t gw_wave_bytes(int fill, int64_t D, int64_t S)
{
    const int64_t W = (S + 1 + 31) / 32, dense = D * (S + 1);
    const size_t f = (size_t)GW_KC + 32 + 256 + 256 + 256 + GR_WSTRIDE + GR_DLINE * 4 + 256;
    const size_t c = (size_t)3 * D * W * 4 + ((dense + 3) & ~(int64_t)3) + 2 * ((D + 3) & ~(int64_t)3);
    return ((size_t)GW_KC + (fill ? f : c) + 15) & ~(size_t)15;
}

template <int FILL>
__global__ void __launch_bounds__(64 * GW_NW) k_gm_tables_w(GmTabArgs a, int wstride)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gtp = xcd_items8(blockIdx.x, gridDim.x) * GW_NW + wave;
    if (gtp >= a.ngtp) return;
    if constexpr (FILL) {
        if (a.ovf && *a.ovf) return;
    }
    char *base = reinterpret_cast<char *>(dyn) + (size_t)wave * wstride;
    uint32_t *kc = reinterpret_cast<uint32_t *>(base);
    char *rest = base + GW_KC;
    const int64_t p = gtp % a.P;
    const int64_t gt = gtp / a.P;
    const int64_t t = gt % a.T;
    const int64_t g = gt / a.T;
    const int64_t row0 = (t * a.P + p) * a.DS;
    const uint32_t S = (uint32_t)a.S;
    const int64_t S1 = a.S + 1;
    const uint32_t S1u = (uint32_t)S1;
    const bool have_passes = FILL && a.voff != nullptr;

    // the lane's nine chain slots: the chain ids first, then every slot's table entry -- independent loads, one round trip
    // each (round 6: as a loop of dependent loads this was 9 x 2 round trips per wavefront)
    constexpr int NIT = (GC_CG + 63) / 64;
    uint32_t cids[NIT];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int i = lane + 64 * k;
        cids[k] = i < GC_CG ? a.order[g * GC_CG + i] : GC_DEAD;
    }
    uint4 rvs[NIT];
    uint32_t cps_in[NIT];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int i = lane + 64 * k;
        rvs[k] = make_uint4(0, 0, 0, 0);
        cps_in[k] = 0;
        if (cids[k] != GC_DEAD) {
            const int64_t e = (((int64_t)cids[k] * a.T + t) * a.P + p) * 4;
            rvs[k] = *reinterpret_cast<const uint4 *>(a.rowoff + e);     // (cc, fc, cf, ff: k_gf_tables)
            if constexpr (FILL)
                if (have_passes) cps_in[k] = a.cpass[gtp * GC_CG + i];
        }
    }
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int i = lane + 64 * k;
        if (i >= GC_CG) continue;
        uint32_t w = 0xffu << 22 | GW_PAD;
        if (cids[k] != GC_DEAD) {
            const uint32_t v0 = rvs[k].x - (uint32_t)row0, v2 = rvs[k].z - (uint32_t)row0;
            const uint32_t dc = v0 / S, sc = v0 % S, df = v2 / S;
            // the floor node of ceil node sc is slot sc of the line (sc = 0: the wrap copy): B = dc (S+1) + sc, A = df (S+1) + sc
            w = ((dc * S1u + sc) << 11) | (df * S1u + sc);
            if constexpr (FILL) w |= cps_in[k] << 22;
        }
        kc[i] = w;
    }

    if constexpr (!FILL) {
        // ---------------- count phase: passes along the duration axis
        const int W = (int)((a.S + 1 + 31) / 32);
        const int64_t dense_n = a.D * S1;
        uint32_t *bC = reinterpret_cast<uint32_t *>(rest), *bF = bC + a.D * W, *bX = bF + a.D * W;
        uint8_t *mark = reinterpret_cast<uint8_t *>(bX + a.D * W);
        uint8_t *linepass = mark + ((dense_n + 3) & ~(int64_t)3);
        uint8_t *linesplit = linepass + ((a.D + 3) & ~(int64_t)3);
        for (int i = lane; i < 3 * a.D * W; i += 64) bC[i] = 0;
        for (int i = lane; i < a.D; i += 64) { linepass[i] = 0; linesplit[i] = 0; }
        wave_sync();
        for (int i = lane; i < GC_CG; i += 64) {
            const uint32_t k22 = kc[i] & GW_PAD;
            if (k22 == GW_PAD) continue;
            const uint32_t sb = k22 >> 11, sa = k22 & 0x7ffu;
            const uint32_t dc = sb / S1u, sc = sb - dc * S1u, df = sa / S1u;
            atomicOr(&bC[dc * W + (sc >> 5)], 1u << (sc & 31));
            atomicOr(&bC[dc * W + ((sc + 1) >> 5)], 1u << ((sc + 1) & 31));
            atomicOr(&bF[df * W + (sc >> 5)], 1u << (sc & 31));
            atomicOr(&bF[df * W + ((sc + 1) >> 5)], 1u << ((sc + 1) & 31));
            atomicOr(&bX[dc * W + (sc >> 5)], 1u << (sc & 31));
        }
        wave_sync();
        int np = 0;
        if (lane == 0) np = gm_count_passes(bC, bF, bX, mark, linepass, linesplit, (int)a.D, W, S, S1, a.cap, a.S);
        wave_sync();
        np = __shfl(np, 0, 64);
        for (int i = lane; i < GC_CG; i += 64) {
            const uint32_t k22 = kc[i] & GW_PAD;
            uint8_t cp = 0xff;
            if (k22 != GW_PAD) {
                const uint32_t sb = k22 >> 11;
                const uint32_t dc = sb / S1u, sc = sb - dc * S1u;
                cp = linesplit[dc] ? mark[dc * S1 + sc] : linepass[dc];
            }
            a.cpass[gtp * GC_CG + i] = cp;
        }
        // (pass ids are bytes: a patch cut into more than 250 passes counts as an overflow of the tables)
        if (lane == 0) a.npass[gtp] = np > 250 ? 0x100000u : (uint32_t)np;
        return;
    } else {
        // ---------------- fill phase
        uint32_t *skp = reinterpret_cast<uint32_t *>(rest);                          // sort word of the chain slot in the pass
        uint8_t *nmw = reinterpret_cast<uint8_t *>(rest + GW_KC);                    // [GC_NCONS] chains of the consumer in the pass
        uint16_t *clist = reinterpret_cast<uint16_t *>(rest + GW_KC + 32);           // compact slot -> dense slot
        uint32_t *bits = reinterpret_cast<uint32_t *>(rest + GW_KC + 32 + 256);      // dense slots of the pass, one word per lane
        uint32_t *wpre = bits + 64;                                                  // set bits in front of the word
        double *wimg = reinterpret_cast<double *>(wpre + 64);                        // [GR_WSTRIDE / 8] a consumer's weight record of the step
        uint32_t *dimg = reinterpret_cast<uint32_t *>(wimg + GR_WSTRIDE / 8);         // [GR_DLINE] its descriptor line
        uint32_t *srt = dimg + GR_DLINE;                                             // [64] the consumer's sort words by walk position
        for (int e = lane; e < GR_WSTRIDE / 8; e += 64) wimg[e] = 0.0;               // (the entries of positions 37..39: never used)
        for (int e = lane; e < GR_DLINE; e += 64) dimg[e] = 0;
        const int npass = have_passes ? (int)a.npass[gtp] : 1;
        const int64_t v0 = a.voff ? (int64_t)a.voff[gtp] : p;
        uint32_t moved = 0;
        auto row_of = [&](uint32_t sl) { const uint32_t d = sl / S1u, s1 = sl % S1u; return d * S + (s1 ? s1 - 1 : S - 1); };
        wave_sync();
        for (int k = 0; k < npass; k++) {
            // dense slots of the pass -> compact slots; the sort words of the pass: (key, pads: all ones) << 6 | slot in the
            // consumer -- unique inside a consumer
            bits[lane] = 0;
            wave_sync();
            for (int i = lane; i < GC_CG; i += 64) {
                const uint32_t w = kc[i];
                const bool mine = (w >> 22) == (uint32_t)k;          // (no chain: pass 0xff)
                skp[i] = (mine ? (w & GW_PAD) : GW_PAD) << 6 | (uint32_t)(i % GC_NCHAIN);
                if (!mine) continue;
                const uint32_t sa = w & 0x7ffu, sb = (w >> 11) & 0x7ffu;
                atomicOr(&bits[sa >> 5], 1u << (sa & 31));
                atomicOr(&bits[(sa + 1) >> 5], 1u << ((sa + 1) & 31));
                atomicOr(&bits[sb >> 5], 1u << (sb & 31));
                atomicOr(&bits[(sb + 1) >> 5], 1u << ((sb + 1) & 31));
            }
            wave_sync();
            const uint32_t word = bits[lane];
            const int cnt = __popc(word);
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            wpre[lane] = (uint32_t)(incl - cnt);
            const int n = __shfl(incl, 63, 64);       // <= cap (count phase)
            moved += (uint32_t)n;
            {
                uint32_t wd = word;
                int q = incl - cnt;
                while (wd) {
                    const int b = __ffs((int)wd) - 1;
                    wd &= wd - 1;
                    if (q < 128) clist[q] = (uint16_t)(lane * 32 + b);
                    q++;
                }
            }
            if (lane < GC_NCONS) {
                int nmem = 0;
                for (int q = 0; q < GC_NCHAIN; q++) nmem += (skp[lane * GC_NCHAIN + q] >> 6) != GW_PAD;
                nmw[lane] = (uint8_t)nmem;
            }
            wave_sync();
            auto cidx = [&](uint32_t x) { return wpre[x >> 5] + (uint32_t)__popc(bits[x >> 5] & ((1u << (x & 31)) - 1u)); };
            // ---- row requests: pairs of compact neighbours whose rows ascend by less than 256, singles otherwise; elements
            // lane and lane + 64
            uint32_t req[2] = {0, 0};
            int rq_idx[2] = {-1, -1};
            int carry_st = 0;
            uint32_t nlead = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = lane + 64 * h;
                const bool in = i < n && i < 128;
                const uint32_t r_i = in ? row_of(clist[i]) : 0u;
                const uint32_t r_p = (in && i > 0) ? row_of(clist[i - 1]) : 0u;
                const bool pairable = in && i > 0 && r_i > r_p && r_i - r_p < 256;   // may follow its predecessor in a pair
                // start of the run of pairable elements the element is in: inclusive max scan of (run start ? i : 0)
                int st = (in && !pairable) ? i : 0;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(st, off, 64);
                    if (lane >= off) st = max(st, o);
                }
                st = max(st, carry_st);
                carry_st = __shfl(st, 63, 64);
                const bool leads = in && (((i - st) & 1) == 0);        // first row of a pair, or a single
                uint32_t r_n = 0;
                bool pair = false;
                if (leads && i + 1 < n) {
                    r_n = row_of(clist[i + 1]);
                    pair = r_n > r_i && r_n - r_i < 256;
                }
                const uint64_t m = __ballot(leads);
                if (leads) {
                    rq_idx[h] = (int)(nlead + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
                    req[h] = r_i | ((pair ? r_n - r_i : 0u) << 16) | ((uint32_t)i << 24);
                    if (rq_idx[h] >= GC_NLOAD * GC_LREQ) rq_idx[h] = -1;     // (cannot happen: the count phase bounds the requests)
                }
                nlead += (uint32_t)__popcll(m);
            }
            const int nreq = min((int)nlead, GC_NLOAD * GC_LREQ);
            for (int iv = 0; iv < a.nvar; iv++) {
                const int64_t s = (v0 + k) * a.nvar + iv;
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (rq_idx[h] >= 0) {
                        const int ll = rq_idx[h] % GC_NLOAD;
                        a.ltab[((gt * (a.smax + 3) + s) * GC_NLOAD + ll) * GC_LTABDW + 2 + rq_idx[h] / GC_NLOAD] = req[h];
                    }
                if (lane < GC_NLOAD) {
                    uint32_t *h = a.ltab + ((gt * (a.smax + 3) + s) * GC_NLOAD + lane) * GC_LTABDW;
                    h[0] = nreq > lane ? (uint32_t)((nreq - lane + GC_NLOAD - 1) / GC_NLOAD) : 0u;
                    h[1] = (uint32_t)(p * a.DS);
                }
            }
            // ---- a consumer at a time, lane <-> chain slot of the consumer: walk position = its sort words below the slot's --
            // the chains of the pass in cell order (ties by slot), the others (pads) behind them in slot order, as
            // k_gm_tables --; "the next chain opens a new cell" from the smallest word above it.  The consumer's weight record
            // and descriptor line are put together in LDS and leave in 16-byte lanes (the records are position-major: written
            // straight from the lanes, every store instruction scattered 8-byte pieces over dozens of lines -- 90 of the
            // kernel's 236 us on configs[3])
            uint32_t cid_n = lane < GC_NCHAIN ? a.order[g * GC_CG + lane] : GC_DEAD;
            for (int w = 0; w < GC_NCONS; w++) {
                const int j = lane;
                const bool act = lane < GC_NCHAIN;
                const uint32_t cid = cid_n;
                if (w + 1 < GC_NCONS) cid_n = act ? a.order[g * GC_CG + (w + 1) * GC_NCHAIN + lane] : GC_DEAD;
                const uint32_t *grp = skp + w * GC_NCHAIN;
                const uint32_t mysk = act ? grp[j] : 0xffffffffu;
                const uint32_t k22 = mysk >> 6;
                const bool mine = act && k22 != GW_PAD;
                // the chain's factors and slips first: the ranking below covers their round trip
                double fr[4] = {0, 0, 0, 0}, slv[4] = {0, 0, 0, 0};
                if (mine) {
                    const int64_t c = (int64_t)cid;
                    const int64_t e = ((c * a.T + t) * a.P + p) * 4;
                    const double2 f01 = *reinterpret_cast<const double2 *>(a.fac + e), f23 = *reinterpret_cast<const double2 *>(a.fac + e + 2);
                    fr[0] = f01.x; fr[1] = f01.y; fr[2] = f23.x; fr[3] = f23.y;
                    for (int iv = 0; iv < a.nvar; iv++)
                        slv[iv] = a.slips[iv].base[c * a.slips[iv].stride + a.slips[iv].off + (t % a.R) * a.P + p];
                }
                int r = 0;
#pragma unroll
                for (int q = 0; q < GC_NCHAIN; q++) r += grp[q] < mysk;
                // the word behind the slot's in the walk: the words by position, one LDS round trip (a running minimum of
                // the larger words inside the loop above cost three more instructions per comparison)
                if (act) srt[r] = mysk;
                wave_sync();
                const uint32_t succ = (act && r + 1 < GC_NCHAIN) ? srt[r + 1] : 0xffffffffu;
                // (a successor that is a pad -- all ones above the key -- or none at all: the walk's last chain opens nothing)
                const bool next_opens = mine && succ != 0xffffffffu && (succ >> 6) != GW_PAD && (succ >> 6) != k22;
                const uint32_t ca = mine ? cidx(k22 & 0x7ffu) : 0u, cb = mine ? cidx(k22 >> 11) : 0u;
                const int nmem_w = nmw[w];
                for (int iv = 0; iv < a.nvar; iv++) {
                    const int64_t s = (v0 + k) * a.nvar + iv;
                    const uint32_t ring = (uint32_t)((s % 3) * a.cap);
                    if (act) {
                        // weights only: a 256-byte record pair serves eight positions, entry e = {weight e of record 2p, of
                        // record 2p + 1}; a pad (a chain of another pass, an empty chain slot): zero weights into the scratch
                        // accumulator, no row reads
                        const double sl = mine ? slv[iv] : 0.0;
                        double *rec = wimg + (r >> 3) * (GR_PAIR / 8) + ((r >> 2) & 1);
                        for (int kk = 0; kk < 4; kk++) rec[(4 * (r & 3) + kk) * 2] = mine ? fr[kk] * sl : 0.0;     // base.py:676-679 x slip
                        uint32_t *dl = dimg + (r < GR_NHALF ? 2 * r : GR_DHALF + 2 * (r - GR_NHALF));
                        dl[0] = (uint32_t)GR_D_BASE | (uint32_t)(mine ? j : GC_SCRATCH) | ((next_opens ? 1u : 0u) << 31);
                        dl[1] = (ring + ca) | ((ring + cb) << 16);
                        // chains of the consumer in this step: the walk leaves the step at the first checkpoint behind them
                        if (j == 0) dimg[GR_D_NCH] = (uint32_t)nmem_w;
                    }
                    wave_sync();
                    char *wdst = a.wtab + ((gt * GC_NCONS + w) * (a.smax + 1) + s) * (int64_t)GR_WSTRIDE;
                    for (int e = lane; e < GR_WSTRIDE / 16; e += 64)
                        *reinterpret_cast<double2 *>(wdst + e * 16) = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(wimg) + e * 16);
                    uint32_t *ddst = a.dtab + ((gt * GC_NCONS + w) * (a.smax + 1) + s) * GR_DLINE;
                    if (lane < GR_DLINE / 4)
                        *reinterpret_cast<uint4 *>(ddst + lane * 4) = *reinterpret_cast<const uint4 *>(dimg + lane * 4);
                    wave_sync();
                }
            }
            wave_sync();
        }
        if (lane == 0) a.ucount[gtp] = moved;
        // the request lines behind the last step of the (group, target) stay empty: the loaders read three steps ahead
        if (p == a.P - 1 && lane < 3 * GC_NLOAD) {
            const int64_t s = (v0 + npass) * a.nvar + lane / GC_NLOAD;
            uint32_t *h = a.ltab + ((gt * (a.smax + 3) + s) * GC_NLOAD + lane % GC_NLOAD) * GC_LTABDW;
            h[0] = 0;
            h[1] = 0;
        }
    }
}

// ---------------------------------------------------------------------------- stacking
struct GcArgs {
    const double *G[4];
    int nvar, ucap, ntile, mode, xcd_order;
    int64_t C, T, P, N, DS, Ttab, rows_per_target, ngroups;
    const int32_t *tslot;         // [T] table slot of a target (nullptr: Ttab == 1 ? 0 : t)
    int64_t smax;                 // steps per (group, target) the tables are strided by
    const uint32_t *nv;           // [group, target] vsteps (nullptr: P)
    const int *ovf;               // nonzero: the tables overflowed, k_gfstack does the work
    const char *wtab;
    const uint32_t *ltab, *order;
    const double *data, *wscalar;
    double *out, *partial;
    const uint32_t *dtab;         // position descriptors, [(g*T+t)][consumer][step 0..smax][GR_DLINE]
    const double *band_w;         // mode 3: [T,N,2] rows (W[i,i], W[i,i+1]) of the bidiagonal whitening operators
    double *edges;                // mode 3: [C*T, ntile, 2] first / last residual of every tile
};

// VAR > 0: timing experiments of tools/gen_gfruns_asm.py (GR_ABLATIONS builds only; wrong results)
template <int NTH, int VAR>
__global__ void __launch_bounds__(1024) k_gfstack_runs(GcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t gsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.ovf && *a.ovf) return;
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
    const int64_t gt = g * a.Ttab + (a.tslot ? (int64_t)a.tslot[t] : (a.Ttab == 1 ? 0 : t));
    const int64_t n0 = (int64_t)tile * 64;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)gsm;
    const uint32_t rb0 = lds0 + GC_PARAM_BYTES;
    const uint32_t nsteps = (uint32_t)((a.nv ? (int64_t)a.nv[gt] : a.P) * a.nvar);
    if (lane == 0) {
        uint32_t *pb = gsm + wave * 32;
        auto put64 = [&](int k, uint64_t x) { pb[k] = (uint32_t)x; pb[k + 1] = (uint32_t)(x >> 32); };
        if (wave < GC_NCONS) {
            put64(GC_P_WP, (uint64_t)(uintptr_t)(a.wtab + ((gt * GC_NCONS + wave) * (a.smax + 1)) * (int64_t)GR_WSTRIDE));
            put64(GC_P_DP, (uint64_t)(uintptr_t)(a.dtab + ((gt * GC_NCONS + wave) * (a.smax + 1)) * (int64_t)GR_DLINE));
            pb[GC_P_RB0] = rb0;
            pb[GC_P_NSTEP] = nsteps;
            pb[GC_P_MODE] = (uint32_t)a.mode;
            put64(GC_P_DATA, (uint64_t)(uintptr_t)(a.data + t * a.N + n0));
            if (a.mode == GF_RESID_BAND1) {
                // (the bidiagonal misfit stores no residual: the words of out / w / ctn carry the tile's edge pair, the band
                // rows of its first sample and "the tile ends the trace")
                put64(GC_P_OUT, (uint64_t)(uintptr_t)(a.edges + (t * a.ntile + tile) * 2));
                pb[GC_P_CTN] = (n0 + 64 >= a.N) ? 1u : 0u;
                put64(GC_P_W, (uint64_t)(uintptr_t)(a.band_w + (t * a.N + n0) * 2));
            } else {
                put64(GC_P_OUT, (uint64_t)(uintptr_t)(a.out + t * a.N + n0));
                pb[GC_P_CTN] = (uint32_t)(a.T * a.N * 8);
                const double wt = a.wscalar ? a.wscalar[t] : 0.0;
                put64(GC_P_W, (uint64_t)__double_as_longlong(wt));
            }
            pb[GC_P_WLDS] = rb0 + (uint32_t)(GC_NCONS * 16 * GC_TPITCH + wave * 1024);
            put64(GC_P_CID, (uint64_t)(uintptr_t)(a.order + g * GC_CG + wave * GC_NCHAIN));
            put64(GC_P_PART, (uint64_t)(uintptr_t)(a.partial + t * a.ntile + tile));
            pb[GC_P_PCS] = (uint32_t)(a.T * a.ntile * 8);
            pb[GC_P_NVALID] = (uint32_t)min((int64_t)64, a.N - n0);
            pb[GC_P_TRB] = rb0 + (uint32_t)(wave * 16 * GC_TPITCH);
        } else {
            const int ll = wave - GC_NCONS;
            put64(GC_PL_LT, (uint64_t)(uintptr_t)(a.ltab + ((gt * (a.smax + 3)) * GC_NLOAD + ll) * GC_LTABDW));
            put64(GC_PL_GROW, (uint64_t)(uintptr_t)(a.G[0] + (t * a.rows_per_target) * a.N + n0));
            pb[GC_PL_ROWB] = (uint32_t)(a.N * 8);
            pb[GC_PL_RB0] = rb0;
            pb[GC_PL_BUFB] = (uint32_t)(a.ucap * 512);      // ucap: slots of a step's row buffer
            pb[GC_PL_NSTEP] = nsteps;
            pb[GC_PL_NLANES] = (uint32_t)min((int64_t)32, (a.N - n0 + 1) / 2);
            // steps cycle through the slip variables' libraries (one variable: the same base thrice)
            pb[GC_PL_NVAR] = (uint32_t)a.nvar;
            put64(GC_PL_G1, (uint64_t)(uintptr_t)(a.G[a.nvar > 1 ? 1 : 0] + (t * a.rows_per_target) * a.N + n0));
            put64(GC_PL_G2, (uint64_t)(uintptr_t)(a.G[a.nvar > 2 ? 2 : 0] + (t * a.rows_per_target) * a.N + n0));
        }
    }
    __syncthreads();
    const uint32_t paddr = lds0 + (uint32_t)(wave * 128);
    if constexpr (VAR == 0) {
        if (wave < GC_NCONS) { GR_CONSUMER_0(paddr); }
        else if (NTH) { GC_LOADER_1(paddr); }
        else { GC_LOADER_0(paddr); }
    }
#if GR_NVARIANT > 1
#define GR_VARIANT(V) if constexpr (VAR == V) { if (wave < GC_NCONS) { GR_CONSUMER_##V(paddr); } else { GC_LOADER_1(paddr); } }
    GR_VARIANT(1) GR_VARIANT(2) GR_VARIANT(3) GR_VARIANT(4) GR_VARIANT(5) GR_VARIANT(6)
#if GR_NVARIANT > 7
    GR_VARIANT(7) GR_VARIANT(8)
#endif
#if GR_NVARIANT > 9
#define GR_VARIANT_NL(V) if constexpr (VAR == V) { if (wave < GC_NCONS) { GR_CONSUMER_##V(paddr); } }
    GR_VARIANT_NL(9) GR_VARIANT_NL(10) GR_VARIANT_NL(11) GR_VARIANT_NL(12)
#if GR_NVARIANT > 13
    GR_VARIANT_NL(13) GR_VARIANT_NL(14) GR_VARIANT_NL(15) GR_VARIANT_NL(16)
#endif
#if GR_NVARIANT > 17
    GR_VARIANT(17)
#endif
#if GR_NVARIANT > 18
    GR_VARIANT(18) GR_VARIANT(19) GR_VARIANT_NL(20) GR_VARIANT_NL(21) GR_VARIANT_NL(22)
#endif
#endif
#endif
}

// Multilinear batches from 192 chains on take the runs kernel whatever the library's (duration x start-time) grid
// (row passes); what rules it out: an odd sample count (16-byte LDS-DMA lanes), more than three slip variables,
// dense-slot / row ids beyond 16 bits or the table kernel's LDS maps, byte offsets beyond 32 bits.
bool gfstack_ml_applicable(const GfStackCall &k)
{
    const GfKnobs &kn = *k.knobs;
    const int knob = GfKnobs::get(kn.gs_ml, -1);   // 0: off, 1: forced also for small batches
    const int gfk = GfKnobs::get(kn.gf_kernel, -1);
    const bool cg_fixed = GfKnobs::set(kn.gs_cg);
    const SeisLib &L = *k.libs[0];
    if (knob == 0 || gfk == 0) return false;
    if (k.interp != BEATAMD_MULTILINEAR || k.nvar < 1 || k.nvar > 3) return false;
    if (L.N % 2 != 0) return false;
    const int64_t DS = L.D * L.S, dense = L.D * (L.S + 1);
    if (DS < 1 || dense > GR_DENSE_MAX || L.D > 255) return false;
    if (L.T * L.N * 8 >= (int64_t)1 << 32 || DS * L.N * 8 >= (int64_t)1 << 32) return false;
    const bool forced = knob == 1;
    if (!forced && k.C < 192) return false;          // small batches: k_gfstack_dma groups of 64..256
    if (!forced && cg_fixed) return false;           // an explicit group size asks for the k_gfstack_dma family
    return true;
}

int launch_gfstack_ml(beatamd_ctx *ctx, const GfStackCall &k, const uint32_t *rowoff, const double *fac, int64_t Ttab,
                      const int **ovf_out)
{
    const SeisLib &L = *k.libs[0];
    const int64_t DS = L.D * L.S, dense = L.D * (L.S + 1);
    const int64_t ngroups = (k.C + GC_CG - 1) / GC_CG;
    const int64_t GT = ngroups * Ttab, GTP = GT * L.P;
    void *p = nullptr;
    *ovf_out = nullptr;
    // row passes: none when every dense slot of a patch (and the row requests they can take) fits a buffer; else the
    // passes are counted on the device and the tables sized for GR_PASS_ALLOC per patch (BEATAMD_GR_CAP: tests)
    const GfKnobs &kn = *k.knobs;
    const int cap = std::min(GR_CAP, std::max(8, GfKnobs::get(kn.gr_cap, GR_CAP)));
    const bool passes = dense > cap || (L.S > 255 ? dense : dense / 2 + 2 * L.D + 1) > GC_NLOAD * GC_LREQ;
    const int64_t vmax = L.P * (passes ? std::max(1, GfKnobs::get(kn.gr_pass_alloc, GR_PASS_ALLOC)) : 1);
    const int64_t smax = vmax * k.nvar;

    GcOrderArgs oa{};
    oa.C = k.C; oa.T = Ttab; oa.P = L.P; oa.S = L.S; oa.rowoff = rowoff;
    // chains that rupture alike share cells patch after patch -> put them into one wavefront (k_gc_order)
    oa.sort = GfKnobs::get(kn.gc_sort, 1) != 0;
    if (GfKnobs::get(kn.gc_keys, 1)) { oa.key[0] = k.order_key[0]; oa.key[1] = k.order_key[1]; }
    BA_TRY(launch_gc_order(ctx, oa, ngroups, kn));   // (reads the row ids of k_gf_tables, launched before this call)

    GmTabArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.C = k.C; ta.T = Ttab; ta.P = L.P; ta.D = L.D; ta.S = L.S; ta.DS = DS;
    ta.nvar = k.nvar; ta.cap = cap;
    ta.vmax = vmax; ta.smax = smax;
    ta.rowoff = rowoff; ta.fac = fac;
    for (int v = 0; v < k.nvar; v++) ta.slips[v] = k.slips[v];
    ta.R = k.patch_split;
    ta.ngtp = GTP;
    ta.order = oa.order;
    BA_TRY(ctx->get_scratch(SL_GC_STREAM, (size_t)GT * GC_NCONS * (smax + 1) * GR_WSTRIDE + 8192, &p));
    ta.wtab = (char *)p;
    BA_TRY(ctx->get_scratch(SL_GC_HDR, (size_t)GT * (smax + 3) * GC_NLOAD * GC_LTAB + 256, &p));
    ta.ltab = (uint32_t *)p;
    // [ucount GTP][npass GTP][voff GTP][nv GT][ovf 1] + cpass bytes
    BA_TRY(ctx->get_scratch(SL_GS_UCOUNT, (size_t)(3 * GTP + GT + 1) * sizeof(uint32_t) + (size_t)GTP * GC_CG, &p));
    ta.ucount = (uint32_t *)p;
    ta.npass = ta.ucount + GTP;
    uint32_t *voff = ta.npass + GTP, *nv = voff + GTP;
    int *ovf = reinterpret_cast<int *>(nv + GT);
    ta.cpass = reinterpret_cast<uint8_t *>(ovf + 1);
    // (the line behind the last step is read ahead, never used)
    BA_TRY(ctx->get_scratch(SL_GC_META, (size_t)GT * GC_NCONS * (smax + 1) * GR_DLINE * sizeof(uint32_t) + 256, &p));
    ta.dtab = (uint32_t *)p;
    {
        ScopedTimer tm(ctx, "grouptables");
        const int W = (int)((L.S + 1 + 31) / 32);
        const size_t lds = (size_t)3 * L.D * W * 4 + 2 * (size_t)((dense + 3) & ~(int64_t)3) + 2 * (size_t)((L.D + 3) & ~(int64_t)3) + 64;
        // one wavefront per patch (k_gm_tables_w) where a patch's dense slots fit one bitset word per lane (BEATAMD_GM_WAVE=0:
        // the workgroup-per-patch kernels: A/B, tests)
        // -- and where there are enough patches to fill the machine with wavefronts (a patch is nine serial rounds
        // there; 400 patches without station shifts: 45 against 30 us, the tutorial grid 112 against 76 us); BEATAMD_GM_WAVE=1
        // forces it
        const int wave_knob = GfKnobs::get(kn.gm_wave, -1);
        const bool per_wave = dense <= GW_DENSE_MAX && cap <= 128 && L.D <= 255 &&
                              (wave_knob == 1 || (wave_knob != 0 && GTP >= 16 * (int64_t)ctx->num_cu));
        const int ws0 = (int)gw_wave_bytes(0, L.D, L.S), ws1 = (int)gw_wave_bytes(1, L.D, L.S);
        const unsigned wgrid = (unsigned)((GTP + GW_NW - 1) / GW_NW);
        if (per_wave) {
            BA_CHECK((size_t)std::max(ws0, ws1) * GW_NW <= 160 * 1024, BEATAMD_EINVAL, "internal: k_gm_tables_w exceeds LDS");
            BA_HIP(hipFuncSetAttribute((const void *)k_gm_tables_w<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ws0 * GW_NW));
            BA_HIP(hipFuncSetAttribute((const void *)k_gm_tables_w<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ws1 * GW_NW));
        }
        if (passes) {
            BA_HIP(hipMemsetAsync(ovf, 0, sizeof(int), ctx->stream));
            if (per_wave) hipLaunchKernelGGL(k_gm_tables_w<0>, dim3(wgrid), dim3(64 * GW_NW), (size_t)ws0 * GW_NW, ctx->stream, ta, ws0);
            else hipLaunchKernelGGL(k_gm_tables<0>, dim3((unsigned)GTP), dim3(GC_TB), lds, ctx->stream, ta);
            hipLaunchKernelGGL(k_gm_scan, dim3((unsigned)GT), dim3(256), 0, ctx->stream, ta.npass, voff, nv, L.P, vmax, ovf);
            ta.voff = voff;
            ta.ovf = ovf;
        }
        if (per_wave) hipLaunchKernelGGL(k_gm_tables_w<1>, dim3(wgrid), dim3(64 * GW_NW), (size_t)ws1 * GW_NW, ctx->stream, ta, ws1);
        else hipLaunchKernelGGL(k_gm_tables<1>, dim3((unsigned)GTP), dim3(GC_TB), lds, ctx->stream, ta);
    }
    BA_HIP(hipGetLastError());

    GcArgs a;
    memset(&a, 0, sizeof(a));
    for (int v = 0; v < k.nvar; v++) a.G[v] = k.libs[v]->g;
    a.nvar = k.nvar; a.ucap = cap;
    a.ntile = (int)((L.N + 63) / 64);
    a.mode = k.mode;
    a.C = k.C; a.T = L.T; a.P = L.P; a.N = L.N; a.DS = DS;
    a.Ttab = Ttab; a.rows_per_target = L.P * DS;
    a.tslot = k.tslot;
    a.ngroups = ngroups; a.smax = smax;
    a.nv = passes ? nv : nullptr;
    a.ovf = passes ? ovf : nullptr;
    a.wtab = ta.wtab; a.ltab = ta.ltab; a.order = oa.order; a.dtab = ta.dtab;
    a.data = k.data; a.wscalar = k.wscalar; a.out = k.out;
    if (k.mode == GF_RESID_SCALAR || k.mode == GF_RESID_BAND1) {
        BA_TRY(ctx->get_scratch(SL_PARTIAL, (size_t)k.C * L.T * a.ntile * sizeof(double), &p));
        a.partial = (double *)p;
    }
    if (k.mode == GF_RESID_BAND1) {
        BA_CHECK(k.band_w && k.quad && k.data, BEATAMD_EINVAL, "gfstack: mode 3 needs band_w, quad, data");
        BA_TRY(ctx->get_scratch(SL_EDGES, (size_t)k.C * L.T * a.ntile * 2 * sizeof(double), &p));
        a.edges = (double *)p;
        a.band_w = k.band_w;
    }
    int64_t nblocks = ngroups * L.T * a.ntile;
    const int order_knob = GfKnobs::get(kn.gs_order, 1);
    a.xcd_order = (ngroups > 1 && order_knob != 0) ? 1 : 0;
    if (a.xcd_order) nblocks = ((L.T * a.ntile + 7) / 8) * 8 * ngroups;
    BA_CHECK(nblocks < (int64_t)0x7fffffff, BEATAMD_EINVAL, "gfstack: batch too large");
    const int nth_knob = GfKnobs::get(kn.gs_nthint, -1);
    const int nth = nth_knob >= 0 ? (nth_knob != 0) : (ngroups == 1);
    const size_t ring = std::max<size_t>((size_t)3 * cap * 512, (size_t)GC_NCONS * 16 * GC_TPITCH + (size_t)GC_NCONS * 1024);
    const size_t lds = GC_PARAM_BYTES + ring;
    BA_CHECK(lds <= 160 * 1024, BEATAMD_EINVAL, "internal: k_gfstack_runs row buffers exceed LDS");
    snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "k_gfstack_runs<%d,%d>", k.mode, nth);
    snprintf(ctx->gf_plan, sizeof(ctx->gf_plan),
             "runs kernel: 518-chain groups, multilinear; %d row slots per LDS buffer (a patch has D*(S+1) = %lld dense slots), %s",
             cap, (long long)dense,
             passes ? "patches that touch more are staged in passes along the duration axis (tables sized for 6 passes per patch; "
                      "beyond that the streaming kernel takes the batch)" : "one pass per patch");
    ctx->gs_ngtp = GTP;
    ctx->gs_trep = (double)L.T / (double)Ttab;
    ctx->gs_N = L.N;
    ctx->gs_cg = GC_CG;
    ctx->gs_nvar = k.nvar;
    ctx->gs_has_passes = passes;
    {
        ScopedTimer tm(ctx, "gfstack");
        void (*kern)(GcArgs) = nth ? k_gfstack_runs<1, 0> : k_gfstack_runs<0, 0>;
#if GR_NVARIANT > 1
        {
            const int var = GfKnobs::get(kn.gr_var, 0);   // timing experiments (GR_ABLATIONS builds; wrong results)
            void (*vk[])(GcArgs) = {kern, k_gfstack_runs<1, 1>, k_gfstack_runs<1, 2>, k_gfstack_runs<1, 3>, k_gfstack_runs<1, 4>,
                                    k_gfstack_runs<1, 5>, k_gfstack_runs<1, 6>,
#if GR_NVARIANT > 7
                                    k_gfstack_runs<1, 7>, k_gfstack_runs<1, 8>,
#endif
#if GR_NVARIANT > 9
                                    k_gfstack_runs<1, 9>, k_gfstack_runs<1, 10>, k_gfstack_runs<1, 11>, k_gfstack_runs<1, 12>,
#endif
#if GR_NVARIANT > 13
                                    k_gfstack_runs<1, 13>, k_gfstack_runs<1, 14>, k_gfstack_runs<1, 15>, k_gfstack_runs<1, 16>,
#endif
#if GR_NVARIANT > 17
                                    k_gfstack_runs<1, 17>,
#endif
#if GR_NVARIANT > 18
                                    k_gfstack_runs<1, 18>, k_gfstack_runs<1, 19>, k_gfstack_runs<1, 20>, k_gfstack_runs<1, 21>, k_gfstack_runs<1, 22>,
#endif
                                    };
            if (var >= 1 && var < GR_NVARIANT) kern = vk[var];
        }
#endif
        BA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(1024), lds, ctx->stream, a);
    }
    BA_HIP(hipGetLastError());
    if (k.mode == GF_RESID_SCALAR) BA_TRY(launch_sum_tiles(ctx, a.partial, k.C * L.T, a.ntile, k.quad, a.ovf, 0));
    if (k.mode == GF_RESID_BAND1)
        BA_TRY(launch_sum_tiles_band1(ctx, a.partial, a.edges, k.band_w, k.C, L.T, L.N, a.ntile, 64, k.quad, a.ovf, 0));
    *ovf_out = a.ovf;
    return BEATAMD_OK;
}

}  // namespace beatamd
