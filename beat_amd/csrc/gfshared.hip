// gfshared.hip -- chain-shared Green's-function stacking for gfx950.
//
// Same arithmetic as k_gfstack (gfstack.hip; reference beat/ffi/base.py:607-709), different
// mapping.  In a batch of C chains the chains of a group choose, for a given (target, patch),
// among at most D*S library rows, and nearby chains choose the SAME row: for BASELINE config 3
// (D*S = 75) 256 chains use ~20 distinct rows per patch.  k_gfstack streams every chain's
// rows (C x 841 MB per batch; duplicates are caught by L2 at best).  Here a workgroup owns
// (chain group, target, 64-sample tile) and every DISTINCT row segment is fetched from HBM once,
// staged in LDS, and applied to all chains of the group that selected it:
//
//   lane  <-> chain (64 chains per wavefront, WAVES <= 8 wavefronts share the staged rows)
//   acc[] <-> the 64 samples of the tile, in registers with static indices (128 VGPRs)
//   per (patch, slip variable) step: the group's distinct rows (k_gf_group_tables) are staged in
//   LDS; each lane then reads ITS row from LDS (per-lane address) and does 64 fp64 FMAs with
//   its own weight.  No cross-lane traffic, no dynamic register indexing, no atomics.
//
// Kernels of this mapping:
//   k_gfstack_ws      (512-chain groups, one row per chain: the bench kernel) 8 consumer + 4 loader wavefronts, ring of three
//                     LDS row buffers, row passes for patches that touch more than 96 distinct rows (k_ws_tables)
//   k_gfstack_dma     (groups of 64 .. 1024 chains, one or four rows per chain) rows by LDS-DMA into two LDS buffers one
//                     step ahead, one barrier per step, hand-issued conflict-free ds_read_b64 (row pitch 65 doubles) or
//                     ds_read_b128 (pitch 66); all global accesses of its loop are asm statements
// (round 1's single-buffer k_gfstack_shared was retired in round 5: what does not fit two row buffers is stacked by the
// row-pass kernels or by k_gfstack)
//
// HBM bytes per batch drop from C x T x P x N x 8 to (distinct rows) x N x 8; the on-chip work
// (LDS reads = algorithmic bytes, fp64 FMAs) is unchanged: the LDS gather is what bounds these
// kernels.  Per chain the patches and rows are accumulated in the same order with
// the same fma() as in k_gfstack: for one slip variable all kernels produce bitwise identical
// synthetics (tests/test_gpu_parity.py).
#include "kernels.hpp"

namespace beatamd {

constexpr int GS_NT_MAX = 64;         // samples per tile (per lane: that many accumulators)
constexpr int GS_NTHINT_DEFAULT = 1;  // non-temporal row requests of single-group batches unless BEATAMD_GS_NTHINT says otherwise
constexpr int GS_WS_DEFAULT = 1;      // wave-specialised kernel for 512-chain groups unless BEATAMD_GS_WS says otherwise

struct GroupTabArgs {
    int nrow, nvar, CG;
    int64_t C, T, P, DS;
    const uint32_t *rowoff;  // [C,T,P,nrow] global row ids (k_gf_tables)
    const double *fac;       // ml: [C,T,P,4]
    ChainVec slips[4];
    int ucap, ustride;
    uint32_t *urows;   // [(g*T+t)*P+p][ustride], padded with the last id
    uint32_t *uent;    // k_gfstack_dma: [(g*T+t)*P+p][waves][ustride/waves][2] = (row id, LDS slot) of
                       // list entry j = wave + k*waves: a wavefront's entries are contiguous
    int nissue;        // wavefronts that stage rows: CG/64 (k_gfstack_dma) or the loaders of k_gfstack_ws
    int windowed;      // slots chosen by LDS bank window (k_gfstack_dma / ds_read_b64), else dense
    int depth;         // windowed: rows per window (LDS holds 32 * depth slots)
    uint32_t *ucount;  // [(g*T+t)*P+p]
    uint32_t *umax;    // max over ucount (atomicMax; zeroed by the launcher)
    uint16_t *slot;    // [((g*T+t)*P+p)*nrow + k][CG]
    double *w;         // nn: [v][(g*P+p)][CG]   ml: [v][((g*T+t)*P+p)*4 + k][CG]
    int64_t w_var_stride;
};

__device__ __forceinline__ void wave_lds_fence()
{
    // the lanes of a wavefront run in lockstep: wait for its LDS operations, no barrier needed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// one workgroup of CG threads per (group, target, patch); thread <-> chain
__global__ void k_gf_group_tables(GroupTabArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    uint32_t *flags = sm;                 // [DS] presence -> slot
    uint32_t *wsum = sm + a.DS;           // [CG] per-thread partial counts
    uint32_t *gmask = wsum + a.CG + 1;    // [DS] windowed: 32-lane groups of the workgroup using the row
    uint32_t *lst = gmask + a.DS;         // [128] windowed: local row index of list entry `pos`
    uint32_t *slt = lst + 128;            // [128] windowed: LDS slot of list entry `pos`
    const int tid = threadIdx.x, CG = a.CG;
    const int64_t gtp = xcd_items8(blockIdx.x, gridDim.x);       // (g*T + t)*P + p
    const int64_t p = gtp % a.P;
    const int64_t gt = gtp / a.P;
    const int64_t t = gt % a.T;
    const int64_t g = gt / a.T;
    const int64_t c = g * CG + tid;
    const bool live = c < a.C;
    const int64_t row0 = (t * a.P + p) * a.DS;

    for (int64_t i = tid; i < a.DS; i += CG) {
        flags[i] = 0;
        if (a.windowed) gmask[i] = 0;
    }
    __syncthreads();
    uint32_t v[4] = {0, 0, 0, 0};
    if (live)
        for (int k = 0; k < a.nrow; k++) {
            v[k] = a.rowoff[((c * a.T + t) * a.P + p) * a.nrow + k] - (uint32_t)row0;
            flags[v[k]] = 1;  // benign race: every writer stores 1
            if (a.windowed) atomicOr(&gmask[v[k]], 1u << (tid >> 5));
        }
    __syncthreads();
    // exclusive scan of flags in row order -> slot numbers (deterministic): chunks of CG flags,
    // rank inside a wavefront by ballot/popcount, wavefront offsets through LDS
    {
        const int lane = tid & 63, wv = tid >> 6, nw = CG >> 6;
        uint32_t run = 0;   // distinct rows before this chunk (same in every thread)
        for (int64_t base = 0; base < a.DS; base += CG) {
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
                a.urows[gtp * a.ustride + pos] = (uint32_t)(row0 + i);
                flags[i] = pos;
                if (a.windowed) lst[pos] = (uint32_t)i;
            }
            run += total;
            __syncthreads();
        }
        if (tid == 0) {
            a.ucount[gtp] = run;
            // (only needed for small groups; a plain read first keeps 25k workgroups off one atomic)
            if (a.umax && run > *(volatile uint32_t *)a.umax) atomicMax(a.umax, run);
            wsum[CG] = run;
        }
    }
    __syncthreads();
    const int total = (int)wsum[CG];
    if (a.windowed && total > 32) {
        // LDS slots by bank window.  k_gfstack_dma reads row `slot` of a lane at slot * (NT+1)
        // doubles with ds_read_b64: the 32 lanes of a lane group hit the two-bank window
        // (slot + sample) mod 32, so rows whose slots differ by 32 collide when ONE lane group reads
        // both (measured with the population of SURVEY 8(d), 39 rows per step on dense slots:
        // SQ_LDS_BANK_CONFLICT = 39 % of the LDS cycles).  Rows sharing a window are therefore
        // chosen so that no 32-lane group uses two of them, as far as the masks allow: rows in
        // order of decreasing number of lane groups using them, each into the window with the
        // least overlap (then the emptiest, then the lowest).  Wavefront 0; lanes 0..31 are the
        // windows.  Results do not depend on the slots, only the LDS read timing does.
        if (tid < 64) {
            // order = rank by (lane groups using the row, descending; list position): every lane
            // ranks its (up to two) rows against all others -- no cross-lane traffic
            const int r0 = tid, r1 = tid + 64;
            const uint32_t m0 = r0 < total ? gmask[lst[r0]] : 0u, m1 = r1 < total ? gmask[lst[r1]] : 0u;
            const int key0 = r0 < total ? ((__popc(m0) << 8) | (255 - r0)) : -1;
            const int key1 = r1 < total ? ((__popc(m1) << 8) | (255 - r1)) : -1;
            int rank0 = 0, rank1 = 0;
            for (int j = 0; j < total; j++) {
                const int kj = (__popc(gmask[lst[j]]) << 8) | (255 - j);
                rank0 += kj > key0;
                rank1 += kj > key1;
            }
            // the 32 most used rows: one window each
            if (r0 < total && rank0 < 32) slt[r0] = (uint32_t)rank0;
            if (r1 < total && rank1 < 32) slt[r1] = (uint32_t)rank1;
            if (r0 < total) wsum[rank0] = (uint32_t)r0;   // rank -> list position (wsum is free here)
            if (r1 < total) wsum[rank1] = (uint32_t)r1;
            wave_lds_fence();
            // window state in lanes 0..31
            uint32_t wmask = 0;
            int wcnt = 0;
            if (tid < 32) {
                wmask = gmask[lst[wsum[tid]]];
                wcnt = 1;
            }
            // the others in rank order: window with the least overlap, then the emptiest, lowest
            for (int rk = 32; rk < total; rk++) {
                const int ridx = (int)wsum[rk];
                const uint32_t mm = gmask[lst[ridx]];
                int cost = (tid < 32 && wcnt < a.depth)
                    ? ((__popc(wmask & mm) << 12) | (wcnt << 6) | tid) : 0x7fffffff;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cost = min(cost, __shfl_xor(cost, off, 64));
                const int w = cost & 63;
                if (tid == w) {
                    slt[ridx] = (uint32_t)(w + 32 * wcnt);
                    wmask |= mm;
                    wcnt++;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < total; i += CG) flags[lst[i]] = slt[i];
        __syncthreads();
    }
    {
        // pad to the stride with the last id: the stacking kernel reads ids unclamped
        const uint32_t last = total > 0 ? a.urows[gtp * a.ustride + total - 1] : (uint32_t)row0;
        for (int i = total + tid; i < a.ustride; i += CG) a.urows[gtp * a.ustride + i] = last;
        // (row id, LDS slot) of every list entry, grouped by the wavefront that stages it (dense
        // slot numbering unless windows were assigned above)
        const int nw = a.nissue, kstr = a.ustride / nw;
        for (int i = tid; i < a.ustride; i += CG) {
            const uint32_t r = (i < total) ? a.urows[gtp * a.ustride + i] : last;
            uint32_t *e = a.uent + ((gtp * nw + (i % nw)) * kstr + (i / nw)) * 2;
            e[0] = r;
            e[1] = (i < total) ? flags[r - (uint32_t)row0] : 0u;
        }
    }
    for (int k = 0; k < a.nrow; k++)
        a.slot[(gtp * a.nrow + k) * CG + tid] = live ? (uint16_t)flags[v[k]] : (uint16_t)0;
    // weights, transposed so that lane <-> chain loads are coalesced
    for (int iv = 0; iv < a.nvar; iv++) {
        double s = 0.0;
        if (live) s = a.slips[iv].base[c * a.slips[iv].stride + a.slips[iv].off + p];
        double *w = a.w + (int64_t)iv * a.w_var_stride;
        if (a.nrow == 1) {
            if (t == 0) w[(g * a.P + p) * CG + tid] = s;
        } else {
            for (int k = 0; k < 4; k++) {
                // base.py:676-679: ((1-st)(1-rt)) * slip etc.; factor product from k_gf_tables
                double f = live ? a.fac[((c * a.T + t) * a.P + p) * 4 + k] : 0.0;
                w[(gtp * 4 + k) * CG + tid] = f * s;
            }
        }
    }
}

struct GsArgs {
    const double *G[4];
    const float *G32[4];   // float copies (k_gfstack_wsp<1>, k_gfstack_dmaf) or nullptr
    int f32pair;           // k_gfstack_dmaf instead of k_gfstack_dma
    int nvar, nrow;
    int64_t C, T, P, N;
    int CG, ucap, ustride, ntile, nt;
    int dma;  // 1: k_gfstack_dma (two LDS row buffers filled by LDS-DMA)
    int ws;            // k_gfstack_ws: 8 consumer + 4 loader wavefronts, three row buffers
    int nthint;        // k_gfstack_ws / k_gfstack_dma: non-temporal row requests
    int xcd_order;     // k_gfstack_dma: chain groups of a (target, tile) share an XCD
    int64_t ngroups;
    // tables per (group, target, patch), or per (group, patch) when the start times do not depend
    // on the target (no station shifts): Ttab = T or 1; the row ids are then those of target 0 and
    // target t reads rows_per_target * t further on
    const int32_t *tslot;    // [T] table slot of a target (nullptr: Ttab == 1 ? 0 : t)
    int64_t Ttab, rows_per_target;
    // twin launches of small groups (launcher): run only if the batch's largest distinct-row count
    // *guard_umax is <= guard_fit (mode 1) / > guard_fit (mode 2); mode 0: always
    const uint32_t *guard_umax;
    int guard_mode, guard_fit;
    // k_gfstack_ws / _wsp: the tables list vsteps (row passes, k_ws_tables): nv[gt] of them per (group, target)
    // (nullptr: P, one per patch), vmax apart
    const uint32_t *nv;
    int64_t vmax;
    // k_gfstack_ws / _wsp: chain of lane `tid` of group g = order[g*512 + tid] (nullptr: g*512 + tid) -- batches of several
    // groups are cut into groups by bisection along the hypocentre keys (launch_chain_members): a piece of the fault per group
    const uint32_t *order;
    const uint32_t *urows, *uent, *ucount;
    const uint16_t *slot;
    const double *w;
    int64_t w_var_stride;
    const double *data, *wscalar;
    double *out, *partial;
    const double *band_w;   // mode 3: [T,N,2] rows of the bidiagonal whitening operator
    double *edges;          // mode 3: [C*T, ntile, 2] first / last residual of every tile
};

typedef double v2d __attribute__((ext_vector_type(2)));

// 8 x ds_read_b128 of consecutive 16-byte units starting OFF bytes behind `addr` (per-lane LDS
// byte address).  Not volatile (see k_gfstack_dma); `tok` is a loop-variant input so that equal
// addresses of different steps are never merged.  The data is valid after lds_wait8.
template <int OFF>
__device__ __forceinline__ void lds_rd8(v2d (&x)[8], uint32_t addr, int tok)
{
    asm("ds_read_b128 %0, %8 offset:%c10\n\t"
        "ds_read_b128 %1, %8 offset:%c10+16\n\t"
        "ds_read_b128 %2, %8 offset:%c10+32\n\t"
        "ds_read_b128 %3, %8 offset:%c10+48\n\t"
        "ds_read_b128 %4, %8 offset:%c10+64\n\t"
        "ds_read_b128 %5, %8 offset:%c10+80\n\t"
        "ds_read_b128 %6, %8 offset:%c10+96\n\t"
        "ds_read_b128 %7, %8 offset:%c10+112"
        : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7])
        : "v"(addr), "s"(tok), "n"(OFF));
}

// wait until at most NLEFT LDS/scalar operations are outstanding; names the 8 destinations
template <int NLEFT>
__device__ __forceinline__ void lds_wait8(v2d (&x)[8])
{
    asm("s_waitcnt lgkmcnt(%c8)"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
        : "n"(NLEFT));
}

// 8 x ds_read_b64 of consecutive doubles (row pitch NT+1 doubles: the 32 lanes of a ds_read_b64
// lane group then hit 32 distinct two-bank windows for up to 32 different rows)
template <int OFF>
__device__ __forceinline__ void lds_rd8_b64(double (&x)[8], uint32_t addr, int tok)
{
    asm("ds_read_b64 %0, %8 offset:%c10\n\t"
        "ds_read_b64 %1, %8 offset:%c10+8\n\t"
        "ds_read_b64 %2, %8 offset:%c10+16\n\t"
        "ds_read_b64 %3, %8 offset:%c10+24\n\t"
        "ds_read_b64 %4, %8 offset:%c10+32\n\t"
        "ds_read_b64 %5, %8 offset:%c10+40\n\t"
        "ds_read_b64 %6, %8 offset:%c10+48\n\t"
        "ds_read_b64 %7, %8 offset:%c10+56"
        : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7])
        : "v"(addr), "s"(tok), "n"(OFF));
}

template <int OFF>
__device__ __forceinline__ void lds_rd8_b32(float (&x)[8], uint32_t addr, int tok)
{
    asm("ds_read_b32 %0, %8 offset:%c10\n\t"
        "ds_read_b32 %1, %8 offset:%c10+4\n\t"
        "ds_read_b32 %2, %8 offset:%c10+8\n\t"
        "ds_read_b32 %3, %8 offset:%c10+12\n\t"
        "ds_read_b32 %4, %8 offset:%c10+16\n\t"
        "ds_read_b32 %5, %8 offset:%c10+20\n\t"
        "ds_read_b32 %6, %8 offset:%c10+24\n\t"
        "ds_read_b32 %7, %8 offset:%c10+28"
        : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7])
        : "v"(addr), "s"(tok), "n"(OFF));
}

typedef float v2f __attribute__((ext_vector_type(2)));

// four pair reads (8 samples of a row): float pairs by ds_read_b64 (SCALE 1), double pairs by ds_read_b128
// (SCALE 2: the offsets double)
template <int SCALE, int OFF, typename PAIR>
__device__ __forceinline__ void lds_rd4_pair(PAIR (&x)[4], uint32_t addr, int tok)
{
    if constexpr (SCALE == 1)
        asm("ds_read_b64 %0, %4 offset:%c6\n\t"
            "ds_read_b64 %1, %4 offset:%c6+8\n\t"
            "ds_read_b64 %2, %4 offset:%c6+16\n\t"
            "ds_read_b64 %3, %4 offset:%c6+24"
            : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3])
            : "v"(addr), "s"(tok), "n"(OFF));
    else
        asm("ds_read_b128 %0, %4 offset:%c6\n\t"
            "ds_read_b128 %1, %4 offset:%c6+16\n\t"
            "ds_read_b128 %2, %4 offset:%c6+32\n\t"
            "ds_read_b128 %3, %4 offset:%c6+48"
            : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3])
            : "v"(addr), "s"(tok), "n"(2 * OFF));
}

template <int NLEFT, typename PAIR>
__device__ __forceinline__ void lds_wait4_pair(PAIR (&x)[4])
{
    asm("s_waitcnt lgkmcnt(%c4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(NLEFT));
}

template <int NLEFT>
__device__ __forceinline__ void lds_wait8_b64(double (&x)[8])
{
    asm("s_waitcnt lgkmcnt(%c8)"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
        : "n"(NLEFT));
}

// ---------------------------------------------------------------------------------------------
// k_gfstack_dma: the same mapping (lane <-> chain, 64 static accumulators, distinct rows staged
// once per workgroup) with the row staging taken off the waves' critical path:
//   * two LDS row buffers; the rows of step s+1 are written by LDS-DMA
//     (global_load_lds_dwordx4, 16 B per lane, no staging VGPRs, no ds_write pass) while the
//     wavefronts run the LDS-gather + FMA phase of step s -- ONE barrier per step;
//   * row ids (scalar) are fetched two steps ahead, the lane's slot/weight one step ahead.
// hipcc does not count asm memory operations, and any vmcnt(N) it emits for loads it does know of
// would under-count and stall on the DMAs: every global access of the loop is therefore an asm
// statement (row DMAs, slot/weight loads), and the only vmcnt wait is the explicit one at the top
// of a step, in the statement that also copies the freshly loaded slot/weight registers (hipcc
// would otherwise copy them before the wait).  The statements are not volatile (a volatile asm is
// a memory clobber for hipcc and turns the scalar row-id loads into waited vector loads);
// sched_barrier(0) pins them between the barrier and the FMA phase, and the destination
// registers of in-flight loads were checked in the ISA to be untouched until their wait.
// NT = 64: 204 VGPRs, 2 wavefronts per SIMD (one 8-wavefront workgroup per CU).  NT = 32: half the
// accumulators, <= 128 VGPRs, 4 wavefronts per SIMD: two 8-wavefront workgroups per CU (one gathers
// while the other sits in its barrier / DMA-issue phase) or one 16-wavefront workgroup = 1024-chain
// groups (every distinct row staged once for twice the chains).
//
// (Three row buffers with requests two steps ahead, and requests issued inside the gather, were
// tried in this kernel and measured slower -- profiles/archive/r2_variants.md; the deeper pipeline lives
// in k_gfstack_ws, where other wavefronts do the issue work.)
template <int WAVES, int NROW, int MODE, int NT, int B64>
__global__ void __launch_bounds__(WAVES * 64)
    __attribute__((amdgpu_waves_per_eu(NT == 64 ? 2 : 4, NT == 64 ? 2 : 4)))
k_gfstack_dma(GsArgs a)
{
    constexpr int GS_NT = NT;
    constexpr int GS_PITCH = B64 ? NT + 1 : NT + 2;
    constexpr int LPR = NT / 2;         // lanes moving one row segment (16 B each)
    // row ids per wavefront fetched ahead (scalar registers): 32 rows per workgroup and step are
    // covered by the unrolled DMA slots, more (rare) go through a loop
    constexpr int KPRE = WAVES >= 4 ? 64 / WAVES : 8;
    extern __shared__ __attribute__((aligned(16))) double xbuf[];  // [2][ucap][GS_PITCH]
    constexpr int CG = WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.guard_mode && ((a.guard_mode == 1) == (*a.guard_umax > (uint32_t)a.guard_fit))) return;
    int tile;
    int64_t t, g;
    if (a.xcd_order) {
        // Workgroups b, b+8, b+16, ... run on the same XCD (round-robin dispatch).  The chain
        // groups of one (target, sample tile) are numbered b = 8*(ngroups*q + g) + x, so they
        // run on one XCD at about the same time and walk the patches in step: the library rows the
        // groups have in common are fetched from HBM once and served from that XCD's L2 to the
        // others.  Scheduling only; results do not depend on it.
        const int64_t b = blockIdx.x;
        const int64_t x = b & 7, q = b >> 3;
        g = q % a.ngroups;
        const int64_t tt = (q / a.ngroups) * 8 + x;   // t * ntile + tile
        if (tt >= a.T * a.ntile) return;              // grid padded to a multiple of 8 per group
        tile = (int)(tt % a.ntile);
        t = tt / a.ntile;
    } else {
        tile = blockIdx.x % a.ntile;
        const int64_t gt0 = blockIdx.x / a.ntile;  // g*T + t
        t = gt0 % a.T;
        g = gt0 / a.T;
    }
    const int64_t slot = a.tslot ? (int64_t)a.tslot[t] : (a.Ttab == 1 ? 0 : t);
    const int64_t gt = g * a.Ttab + slot;                                        // table cell
    const int64_t tbase = (t - slot) * a.rows_per_target * a.N;                  // doubles
    const int64_t c = g * CG + tid;
    const int64_t N = a.N;
    const int64_t n0 = (int64_t)tile * GS_NT;
    const bool dma_lane = (lane < LPR) && (n0 + lane * 2 < N);   // N even (launcher)
    const uint32_t voff = (uint32_t)((n0 + lane * 2) * 8);      // byte offset inside a row
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)xbuf;
    const int bufsz = a.ucap * GS_PITCH;                          // doubles per buffer

    double acc[GS_NT];
#pragma unroll
    for (int i = 0; i < GS_NT; i++) acc[i] = 0.0;
    uint32_t keep = 0;  // results of the DMA statements (always 0): keeps them alive

    const int P = (int)a.P, nvar = a.nvar;
    const int nsteps = P * nvar;
    // steps run patch-major over (patch, variable); (p, iv) of the steps ahead are advanced
    // incrementally (no integer divisions in the loop) and clamp at the last step
    auto advance = [&](int &p, int &iv) {
        if (++iv == nvar) { iv = 0; ++p; }
        if (p >= P) { p = P - 1; iv = nvar - 1; }
    };
    // row j of the step lands at buf*bufsz + j*PITCH; lanes 0..LPR-1 move its NT samples.
    // (scalar address arithmetic kept short: 32 x 32 -> 64 bit products, one exec region per step)
    const uint32_t rowbytes = (uint32_t)(N * 8);
    // `dep` is an ordering token only (not named in the text): a request that lists the destination
    // register of the step's slot load as input cannot be placed in front of that load
    // `tk` chains the requests of a step: every statement passes it through untouched (in/out
    // operand, no instruction), its final value is folded into `keep` once per step -- the
    // statements are not volatile (see above) and must not be dropped.
    auto dma_row = [&](const double *Gv, uint32_t r, uint32_t slotidx, int boff, uint32_t dep, uint32_t &tk) {
        const uint64_t off = (uint64_t)r * (uint64_t)rowbytes;
        const char *rowp = reinterpret_cast<const char *>(Gv) + off;
        const uint32_t dst = lds0 + (uint32_t)(boff * 8) + slotidx * (uint32_t)(GS_PITCH * 8);
        if (a.nthint)   // non-temporal: single-group batches read every row segment exactly once
            asm("s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %2 nt"
                : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst), "v"(dep));
        else
            asm("s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %2"
                : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst), "v"(dep));
    };
    // (row id, LDS slot) of this wavefront's first KPRE list entries: contiguous in memory, one
    // scalar load instruction (the gather's lgkmcnt waits count scalar loads too)
    struct alignas(KPRE * 8) EntBlock { uint32_t v[2 * KPRE]; };
    const int kstr = a.ustride / WAVES;
    // Table addresses: the (group, target) part is loop invariant and hoisted as 64-bit bases; the
    // per-step part is a 32 x 32 bit product added as an unsigned byte offset (scalar loads take it
    // as their offset operand).  Written out by hand: hipcc kept whole 64 x 64 bit index products
    // inside the loop (about 60 scalar instructions per step).
    const char *const cnt_base = reinterpret_cast<const char *>(a.ucount + gt * a.P);
    const char *const ent_base = reinterpret_cast<const char *>(a.uent + ((gt * a.P) * WAVES + wave) * kstr * 2);
    const uint32_t ent_step = (uint32_t)(a.ustride * 8);   // bytes per patch: WAVES * kstr entries of 8 B
    const double *G_a = nullptr;   // library of the step whose list entries are in rid_a / rsl_a
    auto fetch_ids = [&](int p, int iv, int &U, uint32_t (&rid)[KPRE], uint32_t (&rsl)[KPRE]) {
        U = __builtin_amdgcn_readfirstlane(
            (int)*reinterpret_cast<const uint32_t *>(cnt_base + (uint32_t)p * 4u));
        const EntBlock eb = *reinterpret_cast<const EntBlock *>(ent_base + (uint32_t)p * ent_step);
#pragma unroll
        for (int k = 0; k < KPRE; k++) {   // padded: always in bounds
            rid[k] = eb.v[2 * k];
            rsl[k] = eb.v[2 * k + 1];
        }
        // library base pointer of that step: a scalar load from the kernel arguments, issued with
        // the list entries one step before it is needed (selecting among four register pairs by a
        // run-time index costs a branch maze of ~25 scalar instructions per step)
        G_a = a.G[iv] + tbase;
    };
    auto issue_rows_dep = [&](int p, int iv, int boff, int U, const uint32_t (&rid)[KPRE],
                              const uint32_t (&rsl)[KPRE], uint32_t dep) {
        const double *Gv = G_a;
        if (dma_lane) {
            uint32_t tk = 0;
#pragma unroll
            for (int k = 0; k < KPRE; k++)
                if (wave + k * WAVES < U) dma_row(Gv, rid[k], rsl[k], boff, dep, tk);
            if (U > KPRE * WAVES) {   // rare: more distinct rows than the prefetched ids cover
                const uint32_t *ue = reinterpret_cast<const uint32_t *>(ent_base + (uint32_t)p * ent_step);
                for (int k = KPRE; wave + k * WAVES < U; k++) dma_row(Gv, ue[2 * k], ue[2 * k + 1], boff, dep, tk);
            }
            keep |= tk;
        }
    };
    auto issue_rows = [&](int p, int iv, int buf, int U, const uint32_t (&rid)[KPRE],
                          const uint32_t (&rsl)[KPRE]) {
        issue_rows_dep(p, iv, buf * bufsz, U, rid, rsl, voff);   // (any live VGPR: no ordering needed here)
    };
    // the lane's slot and weight of step s: asm loads (hipcc must not count them, see above);
    // valid after the step-top wait statement, which names them
    uint32_t sl_n[NROW];
    double wl_n[NROW];
    const char *const slot_base = reinterpret_cast<const char *>(a.slot + (gt * a.P * NROW) * CG + tid);
    const char *const w_base = reinterpret_cast<const char *>(
        (NROW == 1) ? a.w + (g * a.P) * CG + tid : a.w + (gt * a.P * 4) * CG + tid);
    const uint32_t slot_step = (uint32_t)(NROW * CG * 2);                    // bytes per patch
    const uint32_t w_step = (uint32_t)((NROW == 1 ? 1 : 4) * CG * 8);
    const int64_t w_var_bytes = a.w_var_stride * 8;
    auto fetch_tabs = [&](int p, int iv) {
        const char *sp = slot_base + (uint32_t)p * slot_step;
        const char *wp = w_base + (uint32_t)p * w_step + (nvar == 1 ? (int64_t)0 : (int64_t)iv * w_var_bytes);
#pragma unroll
        for (int k = 0; k < NROW; k++) {
            const uint16_t *ps = reinterpret_cast<const uint16_t *>(sp) + k * CG;
            const double *pw = reinterpret_cast<const double *>(wp) + k * CG;
            asm("global_load_ushort %0, %1, off" : "=v"(sl_n[k]) : "v"(ps));
            asm("global_load_dwordx2 %0, %1, off" : "=v"(wl_n[k]) : "v"(pw));
        }
    };

    int p1 = 0, iv1 = 0;          // step s+1
    advance(p1, iv1);
    int p2 = p1, iv2 = iv1;       // step s+2
    advance(p2, iv2);
    int U_a;
    uint32_t rid_a[KPRE], rsl_a[KPRE];
    fetch_ids(0, 0, U_a, rid_a, rsl_a);
    issue_rows(0, 0, 0, U_a, rid_a, rsl_a);
    fetch_tabs(0, 0);
    fetch_ids(p1, iv1, U_a, rid_a, rsl_a);
    for (int s = 0; s < nsteps; s++) {
        // the tables of this step and (older) the DMA of this step's rows have landed
        __builtin_amdgcn_sched_barrier(0);
        // (the copy into this step's registers is part of the statement: hipcc would otherwise
        // place it in front of the wait and copy registers whose loads are still in flight)
        uint32_t sl[NROW];
        double wl[NROW];
#define BA_WAIT_COPY(NOUT)                                                                      \
        _Pragma("unroll") for (int k = 0; k < NROW; k++)                                        \
            asm("s_waitcnt vmcnt(" #NOUT ")\n\t"                                                 \
                "v_mov_b32 %0, %2\n\t"                                                          \
                "v_mov_b64 %1, %3"                                                              \
                : "=&v"(sl[k]), "=&v"(wl[k]) : "v"(sl_n[k]), "v"(wl_n[k]))
        BA_WAIT_COPY(0);
#undef BA_WAIT_COPY
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // rows of step s visible; everyone has left the FMA phase of step s-1
        __builtin_amdgcn_sched_barrier(0);
        // rows of step s+1 -> the other buffer, then that step's slot/weight and the ids of step s+2
        if (s + 1 < nsteps) issue_rows(p1, iv1, (s + 1) & 1, U_a, rid_a, rsl_a);
        fetch_tabs(p1, iv1);
        fetch_ids(p2, iv2, U_a, rid_a, rsl_a);
        p1 = p2; iv1 = iv2;
        advance(p2, iv2);
        __builtin_amdgcn_sched_barrier(0);
        const int gbuf = (s & 1) * bufsz;   // doubles: the buffer of step s
        // ---- every lane applies ITS rows with ITS weights.  The 2*NT/4 ds_read_b128 of a row
        // are issued by hand in groups of 8, two groups in flight (hipcc keeps 3-4 reads in
        // flight, which leaves the phase bound by LDS latency instead of LDS throughput); the
        // wait statements name the destination registers, so no FMA can move above its wait.
#pragma unroll
        for (int k = 0; k < NROW; k++) {
            const uint32_t xs = lds0 + (uint32_t)((gbuf + (int)sl[k] * GS_PITCH) * 8);
            const double w = wl[k];
            if (B64) {
                constexpr int NG8 = GS_NT / 8;   // groups of 8 reads = 8 samples
                double ya[8], yb[8];
                lds_rd8_b64<0>(ya, xs, s);
                lds_rd8_b64<64>(yb, xs, s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int gq = 0; gq < NG8; gq++) {
                    double(&cur)[8] = (gq & 1) ? yb : ya;
                    if (gq + 1 < NG8) lds_wait8_b64<8>(cur); else lds_wait8_b64<0>(cur);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 8; q++) acc[gq * 8 + q] = fma(cur[q], w, acc[gq * 8 + q]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (gq + 2 < NG8) switch (gq + 2) {
                    case 2: lds_rd8_b64<128>(cur, xs, s); break;
                    case 3: lds_rd8_b64<192>(cur, xs, s); break;
                    case 4: lds_rd8_b64<256>(cur, xs, s); break;
                    case 5: lds_rd8_b64<320>(cur, xs, s); break;
                    case 6: lds_rd8_b64<384>(cur, xs, s); break;
                    case 7: lds_rd8_b64<448>(cur, xs, s); break;
                    default: break;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            constexpr int NG = GS_NT / 16;   // groups of 8 reads = 16 samples
            v2d xa[8], xb[8];
            lds_rd8<0>(xa, xs, s);
            if (NG > 1) lds_rd8<128>(xb, xs, s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gq = 0; gq < NG; gq++) {
                v2d(&cur)[8] = (gq & 1) ? xb : xa;
                if (gq + 1 < NG) lds_wait8<8>(cur); else lds_wait8<0>(cur);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    acc[gq * 16 + 2 * q] = fma(cur[q].x, w, acc[gq * 16 + 2 * q]);
                    acc[gq * 16 + 2 * q + 1] = fma(cur[q].y, w, acc[gq * 16 + 2 * q + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (gq + 2 < NG) {
                    if (gq + 2 == 2) lds_rd8<256>(cur, xs, s);
                    else if (gq + 2 == 3) lds_rd8<384>(cur, xs, s);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        }
    }
    // The slot/weight loads issued in the last step (for the step after the last) are still in
    // flight: wait for them before their registers can be given to anything else.  Without this the
    // epilogue's store addresses were built in those registers and overwritten by the late data
    // (observed as a memory fault with RESID_STORE and >= 29 groups; tools/audit_hidden_loads.py
    // now follows the loop's exit paths as well).
#pragma unroll
    for (int k = 0; k < NROW; k++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(sl_n[k]), "+v"(wl_n[k]));
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: lane = chain c, acc[i] = synthetics[c, t, n0 + i]
    const bool live = (c < a.C) && (keep == 0);
    const int nvalid = (int)min((int64_t)GS_NT, N - n0);
    if (MODE == GF_STORE_SYN) {
        if (live) {
            double *o = a.out + (c * a.T + t) * N + n0;
#pragma unroll
            for (int i = 0; i < GS_NT; i++)
                if (i < nvalid) o[i] = acc[i];
        }
        return;
    }
    __syncthreads();
    if (tid < GS_NT) xbuf[tid] = (tid < nvalid) ? a.data[t * N + n0 + tid] : 0.0;
    __syncthreads();
    if (MODE == GF_RESID_STORE) {
        double *o = a.out + (c * a.T + t) * N + n0;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 8) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i++)
                if (live && i < nvalid) o[i] = xbuf[i] - acc[i];  // seismic.py:1332
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const double w = a.wscalar[t];
        double q = 0.0;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 8) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i++)
                if (i < nvalid) {
                    const double tt = w * (xbuf[i] - acc[i]);
                    q = fma(tt, tt, q);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (live) a.partial[(c * a.T + t) * a.ntile + tile] = q;
    }
}


// k_gfstack_dmaf: k_gfstack_dma on the float copies of the libraries (beatamd_seis_gflib_store_f32), 64-sample
// tiles: 256-byte row segments by full-wave global_load_lds_dword, float rows at a pitch of 33 pairs, float
// PAIRS gathered by ds_read_b64 (half the LDS instructions of the f64 kernel -- the multilinear instance of
// which is bound by exactly those), operands widened in front of the f64 FMA.  Same values, same order:
// bit-identical to the f64 kernels on the (float-representable) library.
template <int WAVES, int NROW, int MODE>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_gfstack_dmaf(GsArgs a)
{
    constexpr int GS_NT = 64;
    constexpr int GS_PITCH = GS_NT + 2;   // floats: 33 pairs
    // row ids per wavefront fetched ahead (scalar registers): 32 rows per workgroup and step are
    // covered by the unrolled DMA slots, more (rare) go through a loop
    constexpr int KPRE = WAVES >= 4 ? 64 / WAVES : 8;
    extern __shared__ __attribute__((aligned(16))) double xbuf[];  // float [2][ucap][GS_PITCH] (+ the epilogue's data tile)
    constexpr int CG = WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.guard_mode && ((a.guard_mode == 1) == (*a.guard_umax > (uint32_t)a.guard_fit))) return;
    int tile;
    int64_t t, g;
    if (a.xcd_order) {
        // Workgroups b, b+8, b+16, ... run on the same XCD (round-robin dispatch).  The chain
        // groups of one (target, sample tile) are numbered b = 8*(ngroups*q + g) + x, so they
        // run on one XCD at about the same time and walk the patches in step: the library rows the
        // groups have in common are fetched from HBM once and served from that XCD's L2 to the
        // others.  Scheduling only; results do not depend on it.
        const int64_t b = blockIdx.x;
        const int64_t x = b & 7, q = b >> 3;
        g = q % a.ngroups;
        const int64_t tt = (q / a.ngroups) * 8 + x;   // t * ntile + tile
        if (tt >= a.T * a.ntile) return;              // grid padded to a multiple of 8 per group
        tile = (int)(tt % a.ntile);
        t = tt / a.ntile;
    } else {
        tile = blockIdx.x % a.ntile;
        const int64_t gt0 = blockIdx.x / a.ntile;  // g*T + t
        t = gt0 % a.T;
        g = gt0 / a.T;
    }
    const int64_t slot = a.tslot ? (int64_t)a.tslot[t] : (a.Ttab == 1 ? 0 : t);
    const int64_t gt = g * a.Ttab + slot;                                        // table cell
    const int64_t tbase = (t - slot) * a.rows_per_target * a.N;                  // doubles
    const int64_t c = g * CG + tid;
    const int64_t N = a.N;
    const int64_t n0 = (int64_t)tile * GS_NT;
    const bool dma_lane = (n0 + lane < N);
    const uint32_t voff = (uint32_t)((n0 + lane) * 4);          // byte offset inside a row
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)xbuf;
    const int bufsz = a.ucap * GS_PITCH;                          // floats per buffer

    double acc[GS_NT];
#pragma unroll
    for (int i = 0; i < GS_NT; i++) acc[i] = 0.0;
    uint32_t keep = 0;  // results of the DMA statements (always 0): keeps them alive

    const int P = (int)a.P, nvar = a.nvar;
    const int nsteps = P * nvar;
    // steps run patch-major over (patch, variable); (p, iv) of the steps ahead are advanced
    // incrementally (no integer divisions in the loop) and clamp at the last step
    auto advance = [&](int &p, int &iv) {
        if (++iv == nvar) { iv = 0; ++p; }
        if (p >= P) { p = P - 1; iv = nvar - 1; }
    };
    // row j of the step lands at buf*bufsz + j*PITCH; lanes 0..LPR-1 move its NT samples.
    // (scalar address arithmetic kept short: 32 x 32 -> 64 bit products, one exec region per step)
    const uint32_t rowbytes = (uint32_t)(N * 4);
    // `dep` is an ordering token only (not named in the text): a request that lists the destination
    // register of the step's slot load as input cannot be placed in front of that load
    // `tk` chains the requests of a step: every statement passes it through untouched (in/out
    // operand, no instruction), its final value is folded into `keep` once per step -- the
    // statements are not volatile (see above) and must not be dropped.
    auto dma_row = [&](const float *Gv, uint32_t r, uint32_t slotidx, int boff, uint32_t dep, uint32_t &tk) {
        const uint64_t off = (uint64_t)r * (uint64_t)rowbytes;
        const char *rowp = reinterpret_cast<const char *>(Gv) + off;
        const uint32_t dst = lds0 + (uint32_t)(boff * 4) + slotidx * (uint32_t)(GS_PITCH * 4);
        if (a.nthint)   // non-temporal: single-group batches read every row segment exactly once
            asm("s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dword %1, %2 nt"
                : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst), "v"(dep));
        else
            asm("s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dword %1, %2"
                : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst), "v"(dep));
    };
    // (row id, LDS slot) of this wavefront's first KPRE list entries: contiguous in memory, one
    // scalar load instruction (the gather's lgkmcnt waits count scalar loads too)
    struct alignas(KPRE * 8) EntBlock { uint32_t v[2 * KPRE]; };
    const int kstr = a.ustride / WAVES;
    // Table addresses: the (group, target) part is loop invariant and hoisted as 64-bit bases; the
    // per-step part is a 32 x 32 bit product added as an unsigned byte offset (scalar loads take it
    // as their offset operand).  Written out by hand: hipcc kept whole 64 x 64 bit index products
    // inside the loop (about 60 scalar instructions per step).
    const char *const cnt_base = reinterpret_cast<const char *>(a.ucount + gt * a.P);
    const char *const ent_base = reinterpret_cast<const char *>(a.uent + ((gt * a.P) * WAVES + wave) * kstr * 2);
    const uint32_t ent_step = (uint32_t)(a.ustride * 8);   // bytes per patch: WAVES * kstr entries of 8 B
    const float *G_a = nullptr;   // library of the step whose list entries are in rid_a / rsl_a
    auto fetch_ids = [&](int p, int iv, int &U, uint32_t (&rid)[KPRE], uint32_t (&rsl)[KPRE]) {
        U = __builtin_amdgcn_readfirstlane(
            (int)*reinterpret_cast<const uint32_t *>(cnt_base + (uint32_t)p * 4u));
        const EntBlock eb = *reinterpret_cast<const EntBlock *>(ent_base + (uint32_t)p * ent_step);
#pragma unroll
        for (int k = 0; k < KPRE; k++) {   // padded: always in bounds
            rid[k] = eb.v[2 * k];
            rsl[k] = eb.v[2 * k + 1];
        }
        // library base pointer of that step: a scalar load from the kernel arguments, issued with
        // the list entries one step before it is needed (selecting among four register pairs by a
        // run-time index costs a branch maze of ~25 scalar instructions per step)
        G_a = a.G32[iv] + tbase;
    };
    auto issue_rows_dep = [&](int p, int iv, int boff, int U, const uint32_t (&rid)[KPRE],
                              const uint32_t (&rsl)[KPRE], uint32_t dep) {
        const float *Gv = G_a;
        if (dma_lane) {
            uint32_t tk = 0;
#pragma unroll
            for (int k = 0; k < KPRE; k++)
                if (wave + k * WAVES < U) dma_row(Gv, rid[k], rsl[k], boff, dep, tk);
            if (U > KPRE * WAVES) {   // rare: more distinct rows than the prefetched ids cover
                const uint32_t *ue = reinterpret_cast<const uint32_t *>(ent_base + (uint32_t)p * ent_step);
                for (int k = KPRE; wave + k * WAVES < U; k++) dma_row(Gv, ue[2 * k], ue[2 * k + 1], boff, dep, tk);
            }
            keep |= tk;
        }
    };
    auto issue_rows = [&](int p, int iv, int buf, int U, const uint32_t (&rid)[KPRE],
                          const uint32_t (&rsl)[KPRE]) {
        issue_rows_dep(p, iv, buf * bufsz, U, rid, rsl, voff);   // (any live VGPR: no ordering needed here)
    };
    // the lane's slot and weight of step s: asm loads (hipcc must not count them, see above);
    // valid after the step-top wait statement, which names them
    uint32_t sl_n[NROW];
    double wl_n[NROW];
    const char *const slot_base = reinterpret_cast<const char *>(a.slot + (gt * a.P * NROW) * CG + tid);
    const char *const w_base = reinterpret_cast<const char *>(
        (NROW == 1) ? a.w + (g * a.P) * CG + tid : a.w + (gt * a.P * 4) * CG + tid);
    const uint32_t slot_step = (uint32_t)(NROW * CG * 2);                    // bytes per patch
    const uint32_t w_step = (uint32_t)((NROW == 1 ? 1 : 4) * CG * 8);
    const int64_t w_var_bytes = a.w_var_stride * 8;
    auto fetch_tabs = [&](int p, int iv) {
        const char *sp = slot_base + (uint32_t)p * slot_step;
        const char *wp = w_base + (uint32_t)p * w_step + (nvar == 1 ? (int64_t)0 : (int64_t)iv * w_var_bytes);
#pragma unroll
        for (int k = 0; k < NROW; k++) {
            const uint16_t *ps = reinterpret_cast<const uint16_t *>(sp) + k * CG;
            const double *pw = reinterpret_cast<const double *>(wp) + k * CG;
            asm("global_load_ushort %0, %1, off" : "=v"(sl_n[k]) : "v"(ps));
            asm("global_load_dwordx2 %0, %1, off" : "=v"(wl_n[k]) : "v"(pw));
        }
    };

    int p1 = 0, iv1 = 0;          // step s+1
    advance(p1, iv1);
    int p2 = p1, iv2 = iv1;       // step s+2
    advance(p2, iv2);
    int U_a;
    uint32_t rid_a[KPRE], rsl_a[KPRE];
    fetch_ids(0, 0, U_a, rid_a, rsl_a);
    issue_rows(0, 0, 0, U_a, rid_a, rsl_a);
    fetch_tabs(0, 0);
    fetch_ids(p1, iv1, U_a, rid_a, rsl_a);
    for (int s = 0; s < nsteps; s++) {
        // the tables of this step and (older) the DMA of this step's rows have landed
        __builtin_amdgcn_sched_barrier(0);
        // (the copy into this step's registers is part of the statement: hipcc would otherwise
        // place it in front of the wait and copy registers whose loads are still in flight)
        uint32_t sl[NROW];
        double wl[NROW];
#define BA_WAIT_COPY(NOUT)                                                                      \
        _Pragma("unroll") for (int k = 0; k < NROW; k++)                                        \
            asm("s_waitcnt vmcnt(" #NOUT ")\n\t"                                                 \
                "v_mov_b32 %0, %2\n\t"                                                          \
                "v_mov_b64 %1, %3"                                                              \
                : "=&v"(sl[k]), "=&v"(wl[k]) : "v"(sl_n[k]), "v"(wl_n[k]))
        BA_WAIT_COPY(0);
#undef BA_WAIT_COPY
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // rows of step s visible; everyone has left the FMA phase of step s-1
        __builtin_amdgcn_sched_barrier(0);
        // rows of step s+1 -> the other buffer, then that step's slot/weight and the ids of step s+2
        if (s + 1 < nsteps) issue_rows(p1, iv1, (s + 1) & 1, U_a, rid_a, rsl_a);
        fetch_tabs(p1, iv1);
        fetch_ids(p2, iv2, U_a, rid_a, rsl_a);
        p1 = p2; iv1 = iv2;
        advance(p2, iv2);
        __builtin_amdgcn_sched_barrier(0);
        const int gbuf = (s & 1) * bufsz;   // doubles: the buffer of step s
        // ---- every lane applies ITS rows with ITS weights.  The 2*NT/4 ds_read_b128 of a row
        // are issued by hand in groups of 8, two groups in flight (hipcc keeps 3-4 reads in
        // flight, which leaves the phase bound by LDS latency instead of LDS throughput); the
        // wait statements name the destination registers, so no FMA can move above its wait.
#pragma unroll
        for (int k = 0; k < NROW; k++) {
            const uint32_t xs = lds0 + (uint32_t)((gbuf + (int)sl[k] * GS_PITCH) * 4);
            const double w = wl[k];
            constexpr int NG = GS_NT / 8;   // groups of 4 pair reads = 8 samples
            v2f ya[4], yb[4];
            lds_rd4_pair<1, 0>(ya, xs, s);
            lds_rd4_pair<1, 32>(yb, xs, s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gq = 0; gq < NG; gq++) {
                v2f(&cur)[4] = (gq & 1) ? yb : ya;
                if (gq + 1 < NG) lds_wait4_pair<4>(cur); else lds_wait4_pair<0>(cur);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    acc[gq * 8 + 2 * q] = fma((double)cur[q].x, w, acc[gq * 8 + 2 * q]);
                    acc[gq * 8 + 2 * q + 1] = fma((double)cur[q].y, w, acc[gq * 8 + 2 * q + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (gq + 2 < NG) switch (gq + 2) {
                case 2: lds_rd4_pair<1, 64>(cur, xs, s); break;
                case 3: lds_rd4_pair<1, 96>(cur, xs, s); break;
                case 4: lds_rd4_pair<1, 128>(cur, xs, s); break;
                case 5: lds_rd4_pair<1, 160>(cur, xs, s); break;
                case 6: lds_rd4_pair<1, 192>(cur, xs, s); break;
                case 7: lds_rd4_pair<1, 224>(cur, xs, s); break;
                default: break;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // The slot/weight loads issued in the last step (for the step after the last) are still in
    // flight: wait for them before their registers can be given to anything else.  Without this the
    // epilogue's store addresses were built in those registers and overwritten by the late data
    // (observed as a memory fault with RESID_STORE and >= 29 groups; tools/audit_hidden_loads.py
    // now follows the loop's exit paths as well).
#pragma unroll
    for (int k = 0; k < NROW; k++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(sl_n[k]), "+v"(wl_n[k]));
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: lane = chain c, acc[i] = synthetics[c, t, n0 + i]
    const bool live = (c < a.C) && (keep == 0);
    const int nvalid = (int)min((int64_t)GS_NT, N - n0);
    if (MODE == GF_STORE_SYN) {
        if (live) {
            double *o = a.out + (c * a.T + t) * N + n0;
#pragma unroll
            for (int i = 0; i < GS_NT; i++)
                if (i < nvalid) o[i] = acc[i];
        }
        return;
    }
    __syncthreads();
    if (tid < GS_NT) xbuf[tid] = (tid < nvalid) ? a.data[t * N + n0 + tid] : 0.0;
    __syncthreads();
    if (MODE == GF_RESID_STORE) {
        double *o = a.out + (c * a.T + t) * N + n0;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 8) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i++)
                if (live && i < nvalid) o[i] = xbuf[i] - acc[i];  // seismic.py:1332
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const double w = a.wscalar[t];
        double q = 0.0;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 8) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i++)
                if (i < nvalid) {
                    const double tt = w * (xbuf[i] - acc[i]);
                    q = fma(tt, tt, q);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (live) a.partial[(c * a.T + t) * a.ntile + tile] = q;
    }
}



// ---------------------------------------------------------------------------------------------
// k_gfstack_ws: wave-specialised form of k_gfstack_dma (512-chain groups).  Ablations of
// k_gfstack_dma on the population of SURVEY 8(d) (same box, ms per launch; timing only): as shipped
// 7.58; row requests replaced by s_nop 6.03; no request block at all 4.96; no LDS reads / FMAs at all
// 6.74 -- a wavefront's step is the SERIAL sum of request issue and gather, and with only one
// step of rows in flight the stream also waits for its own latency (rows two steps ahead without
// the gather: 5.1 ms = the chip's LDS-DMA fill rate).  Here a workgroup is CW = 8 consumer wavefronts
// (lane <-> chain, the accumulators, slot/weight loads, gather + FMA: nothing else) plus LW = 4
// loader wavefronts that issue the row requests two steps ahead into a ring of three LDS row
// buffers.  One s_barrier per step, executed by all twelve wavefronts:
//   loader   : wait until at most its requests of step s+1 are in flight (vmcnt(k), k = what it
//              issued last step, so its rows of step s have landed); barrier(s); issue its
//              share of the rows of step s+2 into buffer (s+2) mod 3 (last read in step s-1)
//   consumer : wait for its slot/weight of step s (hidden asm loads, one step ahead);
//              barrier(s); issue the slot/weight loads of step s+1; gather + FMA from buffer s mod 3
// 12 wavefronts per CU need <= 168 VGPRs (3 per SIMD): tables are addressed with scalar bases +
// 32-bit lane offsets.  Same arithmetic, same order: bitwise equal to the other kernels.
template <int NROW, int MODE, int NB, int NTH>
__global__ void __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_gfstack_ws(GsArgs a)
{
    constexpr int CW = 8, LW = 4;
    constexpr int GS_NT = 64;
    constexpr int GS_PITCH = GS_NT + 1;   // ds_read_b64 layout
    constexpr int LPR = GS_NT / 2;        // lanes moving one row segment (16 B each)
    constexpr int KPRE = 16;              // list entries per loader fetched ahead (64 rows per step)
    extern __shared__ __attribute__((aligned(16))) double xbuf[];  // [3][slots][GS_PITCH]
    constexpr int CG = CW * 64;
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
    const int64_t slot = a.tslot ? (int64_t)a.tslot[t] : (a.Ttab == 1 ? 0 : t);
    const int64_t gt = g * a.Ttab + slot;                                        // table cell
    const int64_t tbase = (t - slot) * a.rows_per_target * a.N;                  // doubles
    const int64_t N = a.N;
    const int64_t n0 = (int64_t)tile * GS_NT;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)xbuf;
    const int bufsz = (a.ucap + 1) * GS_PITCH;                    // doubles per buffer: ucap row slots + the zero row
    // the workgroup's list of vsteps: (patch, row pass) pairs, k_ws_tables -- one per patch unless a patch touches more
    // distinct rows than a buffer holds
    const int P = a.nv ? (int)__builtin_amdgcn_readfirstlane((int)a.nv[gt]) : (int)a.P;
    const int nvar = a.nvar;
    const int nsteps = P * nvar;
    auto advance = [&](int &p, int &iv) {
        if (++iv == nvar) { iv = 0; ++p; }
        if (p >= P) { p = P - 1; iv = nvar - 1; }
    };

    if (wave >= CW) {
        // ==================================== loader ====================================
        const int lw = wave - CW;
        const bool dma_lane = (lane < LPR) && (n0 + lane * 2 < N);   // N even (launcher)
        const uint32_t voff = (uint32_t)((n0 + lane * 2) * 8);      // byte offset inside a row
        const uint32_t rowbytes = (uint32_t)(N * 8);
        uint32_t keep = 0;
        auto dma_row = [&](const double *Gv, uint32_t r, uint32_t slotidx, int boff, uint32_t &tk) {
            const uint64_t off = (uint64_t)r * (uint64_t)rowbytes;
            const char *rowp = reinterpret_cast<const char *>(Gv) + off;
            const uint32_t dst = lds0 + (uint32_t)(boff * 8) + slotidx * (uint32_t)(GS_PITCH * 8);
            // NTH: non-temporal requests -- a row segment is read by this CU once and by no other
            // workgroup when the batch is a single chain group (several groups share rows through L2)
            if (NTH)
                asm("s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, %2 nt"
                    : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst));
            else
                asm("s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, %2"
                    : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst));
        };
        const int kstr = a.ustride / LW;
        const char *const cnt_base = reinterpret_cast<const char *>(a.ucount + gt * a.vmax);
        const char *const ent_base = reinterpret_cast<const char *>(a.uent + ((gt * a.vmax) * LW + lw) * kstr * 2);
        const uint32_t ent_step = (uint32_t)(a.ustride * 8);
        int U_a;
        uint32_t rid[KPRE], rsl[KPRE];
        const double *G_a = nullptr;
        auto fetch_ids = [&](int p, int iv) {
            // constant address space: the tables are written by k_gf_group_tables before this
            // launch, never here, and must come in through scalar loads -- a vector load would be
            // counted in this wavefront's vmcnt together with its row requests
            typedef const __attribute__((address_space(4))) uint32_t *cu32;
            typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
            typedef const __attribute__((address_space(4))) u32x16 *cent;
            U_a = (int)*(cu32)(uintptr_t)(cnt_base + (uint32_t)p * 4u);
            const char *e = ent_base + (uint32_t)p * ent_step;
            const u32x16 e0 = *(cent)(uintptr_t)e;
            const u32x16 e1 = *(cent)(uintptr_t)(e + 64);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                rid[k] = e0[2 * k];      rsl[k] = e0[2 * k + 1];
                rid[8 + k] = e1[2 * k];  rsl[8 + k] = e1[2 * k + 1];
            }
            G_a = a.G[iv] + tbase;
        };
        auto dma_count = [&](int U) { return U > lw ? (U - lw + LW - 1) / LW : 0; };
        // rows lw, lw + LW, ... of the list whose entries are in rid / rsl -> buffer at boff
        auto issue_rows = [&](int p, int boff) {
            if (dma_lane) {
                uint32_t tk = 0;
#pragma unroll
                for (int k = 0; k < KPRE; k++)
                    if (lw + k * LW < U_a) dma_row(G_a, rid[k], rsl[k], boff, tk);
                if (U_a > KPRE * LW) {   // rare: more than 64 distinct rows
                    typedef const __attribute__((address_space(4))) uint32_t *cu32;
                    cu32 ue = (cu32)(uintptr_t)(ent_base + (uint32_t)p * ent_step);
                    for (int k = KPRE; lw + k * LW < U_a; k++) dma_row(G_a, ue[2 * k], ue[2 * k + 1], boff, tk);
                }
                keep |= tk;
            }
        };
        // steps 0 .. NB-2 go out before the loop, then step s+NB-1 behind the barrier of step s
        int boff[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) boff[j] = j * bufsz;   // boff[j]: buffer of step s+j
        int cnt[NB];                                          // cnt[j]: this loader's requests of step s+j
#pragma unroll
        for (int j = 0; j < NB; j++) cnt[j] = 0;
        int pn = 0, ivn = 0;                                  // next step to fetch ids for
        fetch_ids(pn, ivn);
#pragma unroll
        for (int j = 0; j < NB - 1; j++) {
            if (j < nsteps) {
                issue_rows(pn, boff[j]);
                cnt[j] = dma_count(U_a);
            }
            advance(pn, ivn);
            fetch_ids(pn, ivn);                               // ids of step j+1
        }
        int p_i = pn;                                         // step whose ids are in rid / rsl (s+NB-1)
        for (int s = 0; s < nsteps; s++) {
            __builtin_amdgcn_sched_barrier(0);
            // this loader's rows of step s have landed when at most its requests of the NB-2
            // younger steps stay in flight (s_waitcnt takes an immediate; an over-wait is safe)
            {
                int young = 0;
#pragma unroll
                for (int j = 1; j < NB - 1; j++) young += cnt[j];
                uint32_t tk = 0;
#define BA_LWAIT(NOUT) asm("s_waitcnt vmcnt(" #NOUT ")" : "+s"(tk) : "s"(s))
                switch (young) {
                case 0: BA_LWAIT(0); break;   case 1: BA_LWAIT(1); break;   case 2: BA_LWAIT(2); break;
                case 3: BA_LWAIT(3); break;   case 4: BA_LWAIT(4); break;   case 5: BA_LWAIT(5); break;
                case 6: BA_LWAIT(6); break;   case 7: BA_LWAIT(7); break;   case 8: BA_LWAIT(8); break;
                case 9: BA_LWAIT(9); break;   case 10: BA_LWAIT(10); break; case 11: BA_LWAIT(11); break;
                case 12: BA_LWAIT(12); break; case 13: BA_LWAIT(13); break; case 14: BA_LWAIT(14); break;
                case 15: BA_LWAIT(15); break; case 16: BA_LWAIT(16); break; case 17: BA_LWAIT(17); break;
                case 18: BA_LWAIT(18); break; case 19: BA_LWAIT(19); break; case 20: BA_LWAIT(20); break;
                case 21: BA_LWAIT(21); break; case 22: BA_LWAIT(22); break; case 23: BA_LWAIT(23); break;
                case 24: BA_LWAIT(24); break; case 25: BA_LWAIT(25); break; case 26: BA_LWAIT(26); break;
                case 27: BA_LWAIT(27); break; case 28: BA_LWAIT(28); break; case 29: BA_LWAIT(29); break;
                case 30: BA_LWAIT(30); break; case 31: BA_LWAIT(31); break; default: BA_LWAIT(32); break;
                }
#undef BA_LWAIT
                keep |= tk;
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();   // rows of step s published; buffer of step s-1 is free
            __builtin_amdgcn_sched_barrier(0);
            int k_new = 0;
            if (s + NB - 1 < nsteps) {
                issue_rows(p_i, boff[NB - 1]);
                k_new = dma_count(U_a);
            }
            advance(pn, ivn);
            p_i = pn;
            fetch_ids(pn, ivn);
            const int b0 = boff[0];
#pragma unroll
            for (int j = 0; j < NB - 1; j++) { boff[j] = boff[j + 1]; cnt[j] = cnt[j + 1]; }
            boff[NB - 1] = b0;
            cnt[NB - 1] = 0;
            cnt[NB - 2] = k_new;
        }
        __builtin_amdgcn_s_barrier();   // the consumers' barrier "of step nsteps" (their pipelined loop)
        // the consumers' epilogue passes the data tile through LDS behind two barriers
        if (MODE != GF_STORE_SYN) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        if (keep != 0) a.out[0] = 0.0;   // never true: keeps the request statements alive
        return;
    }

    // ==================================== consumer ====================================
    static_assert(NROW == 1, "the loader / consumer kernel exists for single-row interpolation");
    const int64_t c = a.order ? (int64_t)a.order[g * CG + tid] : g * CG + tid;     // (beyond the batch: >= C)
    // the zero row of every buffer (slot index ucap; the loaders never write it): what a lane reads, with weight 0, in
    // the row passes of a patch that do not hold its chain's row
    for (int i = tid; i < NB * GS_PITCH; i += CG) xbuf[(i / GS_PITCH) * bufsz + a.ucap * GS_PITCH + (i % GS_PITCH)] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (visible behind the barrier of step 0)
    double acc[GS_NT];
#pragma unroll
    for (int i = 0; i < GS_NT; i++) acc[i] = 0.0;
    // the lane's slot and weight, fetched one step ahead by asm loads hipcc does not count
    uint32_t sl_n;
    double wl_n;
    // scalar base + 32-bit lane offset: no 64-bit pointers in VGPRs
    const uint32_t voff_s = (uint32_t)tid * 2u;   // (the weight's offset is rebuilt from it per load: one
                                                   // register less is what keeps the step loop free of spills)
    const char *const slot_base = reinterpret_cast<const char *>(a.slot + (gt * a.vmax) * CG);
    const char *const w_base = reinterpret_cast<const char *>(a.w + (gt * a.vmax) * CG);
    const uint32_t slot_step = (uint32_t)(CG * 2);
    const uint32_t w_step = (uint32_t)(CG * 8);
    const int64_t w_var_bytes = a.w_var_stride * 8;
    auto tab_slot = [&](int p) { return slot_base + (uint32_t)p * slot_step; };
    auto tab_w = [&](int p, int iv) {
        return w_base + (uint32_t)p * w_step + (nvar == 1 ? (int64_t)0 : (int64_t)iv * w_var_bytes);
    };
    auto fetch_tabs = [&](const char *ps, const char *pw) {
        asm("global_load_ushort %0, %1, %2" : "=v"(sl_n) : "v"(voff_s), "s"(ps));
        uint32_t voff_w;
        asm("v_lshlrev_b32 %1, 2, %2\n\t"
            "global_load_dwordx2 %0, %1, %3" : "=v"(wl_n), "=&v"(voff_w) : "v"(voff_s), "s"(pw));
    };
    // The landed slot / weight are consumed INSIDE asm statements placed behind the wait: a tied
    // ("+v") wait statement lets hipcc copy the still-in-flight register in front of it.
    // xs = LDS byte address of the lane's row = base + slot * (GS_PITCH * 8), GS_PITCH = 65
    auto row_address = [&](uint32_t base) {
        uint32_t x;
        asm("s_waitcnt vmcnt(0)\n\t"
            "v_lshl_add_u32 %0, %1, 6, %1\n\t"
            "v_lshl_add_u32 %0, %0, 3, %2" : "=&v"(x) : "v"(sl_n), "s"(base), "v"(wl_n));
        return x;
    };
    auto landed_weight = [&](uint32_t after) {   // `after`: the row address, i.e. behind the wait
        double x;
        asm("v_mov_b64 %0, %1" : "=v"(x) : "v"(wl_n), "v"(after));
        return x;
    };
    static_assert(GS_PITCH == 65, "row_address multiplies by 65");
    constexpr int NG8 = GS_NT / 8;
    static_assert(NG8 == 8, "the read schedule below is written for 64-sample tiles");
    double ya[8], yb[8];
    int p1 = 0, iv1 = 0;          // position of the step whose slot/weight are fetched next
    advance(p1, iv1);
    int gbuf = 0;                 // row buffer of the step being gathered
    const int ring = NB * bufsz;
    // The gather of a step is 8 groups of 8 ds_read_b64 + 8 FMA, two groups in flight.  It is
    // pipelined across the step boundary: once the reads of step s are all back (before the FMAs
    // of its last group) the wavefront passes the barrier of step s+1 and has that step's first
    // group in flight while it finishes step s -- the LDS pipe does not drain at every barrier.
    double w;
    uint32_t xs;
    fetch_tabs(tab_slot(0), tab_w(0, 0));
    xs = row_address(lds0);
    w = landed_weight(xs);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();     // rows of step 0 visible
    __builtin_amdgcn_sched_barrier(0);
    lds_rd8_b64<0>(ya, xs, -1);
    lds_rd8_b64<64>(yb, xs, -1);
    __builtin_amdgcn_sched_barrier(0);
    fetch_tabs(tab_slot(p1), tab_w(p1, iv1));
    advance(p1, iv1);
    // The tables of step s+2 as RUNNING pointers (round 4).  The loop is sensitive to every scalar instruction -- 28
    // dummy s_add_u32 per step cost 5 % -- and rebuilding both pointers from (patch, variable) with the clamp at the last
    // step took ~25 of them.  Now: five scalar instructions pick the increments (hand-written: hipcc turns the same
    // selects into a dozen), four add them, behind the fetch at the end of the step.  No clamp: the pointers run up to two
    // steps past the last one, into padding the launcher allocates (values fetched there are never used).
    const char *ps_run = tab_slot(p1), *pw_run = tab_w(p1, iv1);
    int iv_run = iv1;                                                                     // variable the pointers stand at
    const int64_t dw_next = (nvar == 1) ? (int64_t)w_step : w_var_bytes;                  // next variable of the patch
    const int64_t dw_wrap = (int64_t)w_step - (int64_t)(nvar - 1) * w_var_bytes;          // first variable, next patch
#pragma clang loop unroll(disable)
    for (int s = 0; s < nsteps; s++) {
        // all scalar bookkeeping of the step first: hipcc sinks FMAs into any block that follows the
        // gather, so nothing below may branch
        int gnext = gbuf + bufsz;
        if (gnext == ring) gnext = 0;
        __builtin_amdgcn_sched_barrier(0);
        // One asm statement per gather group: wait for the group's reads, its 8 FMAs, and the reads
        // of the group after next into the registers just consumed (v_fmac_f64 d, x, w = fma(x, w, d)).
#define BA_FMA8 \
    "v_fmac_f64 %0, %8, %16\n\tv_fmac_f64 %1, %9, %16\n\tv_fmac_f64 %2, %10, %16\n\tv_fmac_f64 %3, %11, %16\n\t" \
    "v_fmac_f64 %4, %12, %16\n\tv_fmac_f64 %5, %13, %16\n\tv_fmac_f64 %6, %14, %16\n\tv_fmac_f64 %7, %15, %16\n\t"
#define BA_RD8(OFF) \
    "ds_read_b64 %8, %17 offset:" #OFF "\n\tds_read_b64 %9, %17 offset:" #OFF "+8\n\t" \
    "ds_read_b64 %10, %17 offset:" #OFF "+16\n\tds_read_b64 %11, %17 offset:" #OFF "+24\n\t" \
    "ds_read_b64 %12, %17 offset:" #OFF "+32\n\tds_read_b64 %13, %17 offset:" #OFF "+40\n\t" \
    "ds_read_b64 %14, %17 offset:" #OFF "+48\n\tds_read_b64 %15, %17 offset:" #OFF "+56"
#define BA_ACC8(G) \
    "+v"(acc[G * 8]), "+v"(acc[G * 8 + 1]), "+v"(acc[G * 8 + 2]), "+v"(acc[G * 8 + 3]), "+v"(acc[G * 8 + 4]), \
        "+v"(acc[G * 8 + 5]), "+v"(acc[G * 8 + 6]), "+v"(acc[G * 8 + 7])
#define BA_Y8(Y) "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]), "+v"(Y[4]), "+v"(Y[5]), "+v"(Y[6]), "+v"(Y[7])
#define BA_FMA4A \
    "v_fmac_f64 %0, %8, %16\n\tv_fmac_f64 %1, %9, %16\n\tv_fmac_f64 %2, %10, %16\n\tv_fmac_f64 %3, %11, %16\n\t"
#define BA_FMA4B \
    "v_fmac_f64 %4, %12, %16\n\tv_fmac_f64 %5, %13, %16\n\tv_fmac_f64 %6, %14, %16\n\tv_fmac_f64 %7, %15, %16\n\t"
#define BA_RD4A(OFF) \
    "ds_read_b64 %8, %17 offset:" #OFF "\n\tds_read_b64 %9, %17 offset:" #OFF "+8\n\t" \
    "ds_read_b64 %10, %17 offset:" #OFF "+16\n\tds_read_b64 %11, %17 offset:" #OFF "+24\n\t"
#define BA_RD4B(OFF) \
    "ds_read_b64 %12, %17 offset:" #OFF "+32\n\tds_read_b64 %13, %17 offset:" #OFF "+40\n\t" \
    "ds_read_b64 %14, %17 offset:" #OFF "+48\n\tds_read_b64 %15, %17 offset:" #OFF "+56"
// in half groups of 4 (LDS operations return in order): 12..16 reads stay in flight instead of
// 8..16 -- measured 0.5 % (39 rows per step) to 1.3 % (21 rows) per launch
#define BA_GROUP(G, Y, OFFNEXT) \
    asm("s_waitcnt lgkmcnt(12)\n\t" BA_FMA4A BA_RD4A(OFFNEXT) "s_waitcnt lgkmcnt(12)\n\t" BA_FMA4B BA_RD4B(OFFNEXT) \
        : BA_ACC8(G), BA_Y8(Y) : "v"(w), "v"(xs))
        BA_GROUP(0, ya, 128);
        BA_GROUP(1, yb, 192);
        BA_GROUP(2, ya, 256);
        BA_GROUP(3, yb, 320);
        BA_GROUP(4, ya, 384);
        BA_GROUP(5, yb, 448);
        __builtin_amdgcn_sched_barrier(0);
        // group 6: every read of this step is back once both groups in flight have landed
        // (yb is named so that group 7 stays behind this wait)
        asm("s_waitcnt lgkmcnt(0)\n\t" BA_FMA8
            : BA_ACC8(6), BA_Y8(ya) : "v"(w), "v"(xs), "v"(yb[0]), "v"(yb[1]), "v"(yb[2]), "v"(yb[3]),
              "v"(yb[4]), "v"(yb[5]), "v"(yb[6]), "v"(yb[7]));
        __builtin_amdgcn_sched_barrier(0);
        gbuf = gnext;
        const uint32_t xs_n = row_address(lds0 + (uint32_t)(gbuf * 8));   // slot / weight of step s+1 have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // rows of step s+1 visible; this wavefront is done with buffer s
        __builtin_amdgcn_sched_barrier(0);
        lds_rd8_b64<0>(ya, xs_n, s);    // first group of step s+1 (after the last step: unused rows)
        __builtin_amdgcn_sched_barrier(0);
        // group 7 with the weight of step s, then the weight of step s+1 and that step's second group
        asm(BA_FMA8
            "v_mov_b64 %16, %18\n\t" BA_RD8(64)
            : BA_ACC8(7), BA_Y8(yb), "+v"(w) : "v"(xs_n), "v"(wl_n));
        xs = xs_n;
        __builtin_amdgcn_sched_barrier(0);
#undef BA_GROUP
#undef BA_RD4B
#undef BA_RD4A
#undef BA_FMA4B
#undef BA_FMA4A
#undef BA_Y8
#undef BA_ACC8
#undef BA_RD8
#undef BA_FMA8
        fetch_tabs(ps_run, pw_run);     // slot / weight of step s+2
        {
            // patch-major over (patch, variable): the slot table moves on with the patch, the weight table every step
            uint32_t ds;
            int64_t dw;
            asm("s_add_i32 %0, %0, 1\n\t"
                "s_cmp_eq_u32 %0, %3\n\t"
                "s_cselect_b32 %0, 0, %0\n\t"
                "s_cselect_b32 %1, %4, 0\n\t"
                "s_cselect_b64 %2, %5, %6"
                : "+s"(iv_run), "=&s"(ds), "=&s"(dw) : "s"(nvar), "s"(slot_step), "s"(dw_wrap), "s"(dw_next) : "scc");
            ps_run += ds;
            pw_run += dw;
        }
    }
    // drain the reads issued for the step after the last (their registers stay reserved until here)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]), "+v"(ya[3]), "+v"(ya[4]), "+v"(ya[5]), "+v"(ya[6]),
                   "+v"(ya[7]), "+v"(yb[0]), "+v"(yb[1]), "+v"(yb[2]), "+v"(yb[3]), "+v"(yb[4]), "+v"(yb[5]),
                   "+v"(yb[6]), "+v"(yb[7]), "+v"(sl_n), "+v"(wl_n));
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: lane = chain c, acc[i] = synthetics[c, t, n0 + i]
    const bool live = (c < a.C);
    const int nvalid = (int)min((int64_t)GS_NT, N - n0);
    if (MODE == GF_STORE_SYN) {
        if (live) {
            double *o = a.out + (c * a.T + t) * N + n0;
#pragma unroll
            for (int i = 0; i < GS_NT; i++)
                if (i < nvalid) o[i] = acc[i];
        }
        return;
    }
    __builtin_amdgcn_s_barrier();
    if (tid < GS_NT) xbuf[tid] = (tid < nvalid) ? a.data[t * N + n0 + tid] : 0.0;
    if (MODE == GF_RESID_BAND1 && tid >= GS_NT && tid < 3 * GS_NT)     // (w0_i, w1_i) of the tile's samples behind the data
        xbuf[tid] = ((tid - GS_NT) >> 1) < nvalid ? a.band_w[(t * N + n0) * 2 + (tid - GS_NT)] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (MODE == GF_RESID_BAND1) {
        // distributions.py:119-138 with a bidiagonal W: y_i = W[i,i] r_i + W[i,i+1] r_{i+1} -- the two products of
        // k_quadform_banded<1> in its order -- for the samples whose neighbour this lane holds; the tile's LAST sample needs
        // the next tile's first residual: both go to `edges`, k_sum_tiles_band1 adds that term (the trace's very last
        // sample has no neighbour and is finished here)
        const double *wbt = xbuf + GS_NT;
        const bool trace_end = n0 + nvalid == N;
        double q = 0.0, ri = xbuf[0] - acc[0];
        const double rfirst = ri;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 8) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i++) {
                if (i + 1 < nvalid) {
                    const double rn = xbuf[i + 1] - acc[i + 1];     // seismic.py:1332
                    double y = fma(wbt[2 * i], ri, 0.0);
                    y = fma(wbt[2 * i + 1], rn, y);
                    q = fma(y, y, q);
                    ri = rn;
                } else if (i + 1 == nvalid && trace_end) {
                    const double y = fma(wbt[2 * i], ri, 0.0);
                    q = fma(y, y, q);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (live) {
            const int64_t e = (c * a.T + t) * a.ntile + tile;
            a.partial[e] = q;
            a.edges[2 * e] = rfirst;
            a.edges[2 * e + 1] = ri;        // residual of the tile's last valid sample
        }
    } else if (MODE == GF_RESID_STORE) {
        // A lane holds 64 consecutive samples of ITS chain: stored directly, every store instruction scatters 64 x 8 bytes
        // over 64 rows of the residual matrix (T*N*8 bytes apart) -- 1 GB that way costs 0.8 ms of a 7 ms launch.  Through a
        // per-wavefront LDS tile [64 chains][16 samples] (pitch 17) the same values leave as 128-byte runs: lane = (chain
        // 4k + lane / 16, sample lane % 16), four full lines per instruction.  The row ring is free (barrier above).
        constexpr int TP = 17;
        double *tw = xbuf + GS_NT + wave * (64 * TP);
        const uint32_t cu = (uint32_t)min(c, (int64_t)0xffffffff);
        const int sl = lane & 15;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) tw[lane * TP + i] = xbuf[i0 + i] - acc[i0 + i];  // seismic.py:1332
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const double v = tw[(k * 4 + (lane >> 4)) * TP + sl];
                const int64_t cc = (int64_t)(uint32_t)__shfl((int)cu, k * 4 + (lane >> 4));   // chain of the row this lane stores
                if (cc < a.C && i0 + sl < nvalid) a.out[(cc * a.T + t) * N + n0 + i0 + sl] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        const double w = a.wscalar[t];
        double q = 0.0;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 8) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i++)
                if (i < nvalid) {
                    const double tt = w * (xbuf[i] - acc[i]);
                    q = fma(tt, tt, q);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (live) a.partial[(c * a.T + t) * a.ntile + tile] = q;
    }
}


// k_gfstack_ws32: k_gfstack_ws on the float copy of the library (beatamd_seis_gflib_store_f32): a row
// segment of a 64-sample tile is 256 bytes -- one full-wave global_load_lds_dword --, the LDS rows hold
// floats (pitch 66 dwords = 33 qwords: conflict-free ds_read_b64 gather of float PAIRS, half the LDS
// instructions of the f64 kernel), every operand is widened by v_cvt_f64_f32 in front of its FMA; accumulation, weights and epilogues stay f64.  The f64 library holds the same
// (float-representable) values, so the result is bit for bit what the f64 kernels give.
// F32 = 0: the same pair gather on the float64 library (A/B: BEATAMD_GS_PAIR=1): double PAIRS by
// ds_read_b128 at a pitch of 66 doubles = 33 x 16 bytes.
template <int F32, int MODE, int NB, int NTH>
__global__ void __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_gfstack_wsp(GsArgs a)
{
    constexpr int CW = 8, LW = 4;
    constexpr int GS_NT = 64;
    constexpr int GS_PITCH = GS_NT + 2;   // elements: 33 pairs, the conflict-free layout of the pair gather
    constexpr int ES = F32 ? 4 : 8;       // bytes per element
    typedef typename std::conditional<F32 != 0, float, double>::type elem_t;
    typedef typename std::conditional<F32 != 0, v2f, v2d>::type pair_t;
    constexpr int KPRE = 16;              // list entries per loader fetched ahead (64 rows per step)
    extern __shared__ __attribute__((aligned(16))) double xbuf[];  // float [3][slots][GS_PITCH] (+ the epilogue's data tile)
    constexpr int CG = CW * 64;
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
    const int64_t slot = a.tslot ? (int64_t)a.tslot[t] : (a.Ttab == 1 ? 0 : t);
    const int64_t gt = g * a.Ttab + slot;                                        // table cell
    const int64_t tbase = (t - slot) * a.rows_per_target * a.N;                  // doubles
    const int64_t N = a.N;
    const int64_t n0 = (int64_t)tile * GS_NT;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)xbuf;
    const int bufsz = (a.ucap + 1) * GS_PITCH;                    // elements per buffer: ucap row slots + the zero row
    // the workgroup's list of vsteps: (patch, row pass) pairs, k_ws_tables -- one per patch unless a patch touches more
    // distinct rows than a buffer holds
    const int P = a.nv ? (int)__builtin_amdgcn_readfirstlane((int)a.nv[gt]) : (int)a.P;
    const int nvar = a.nvar;
    const int nsteps = P * nvar;
    auto advance = [&](int &p, int &iv) {
        if (++iv == nvar) { iv = 0; ++p; }
        if (p >= P) { p = P - 1; iv = nvar - 1; }
    };

    if (wave >= CW) {
        // ==================================== loader ====================================
        const int lw = wave - CW;
        const bool dma_lane = F32 ? (n0 + lane < N) : ((lane < 32) && (n0 + lane * 2 < N));
        const uint32_t voff = F32 ? (uint32_t)((n0 + lane) * 4) : (uint32_t)((n0 + lane * 2) * 8);   // byte offset inside a row
        const uint32_t rowbytes = (uint32_t)(N * ES);
        uint32_t keep = 0;
        auto dma_row = [&](const elem_t *Gv, uint32_t r, uint32_t slotidx, int boff, uint32_t &tk) {
            const uint64_t off = (uint64_t)r * (uint64_t)rowbytes;
            const char *rowp = reinterpret_cast<const char *>(Gv) + off;
            const uint32_t dst = lds0 + (uint32_t)(boff * ES) + slotidx * (uint32_t)(GS_PITCH * ES);
            // NTH: non-temporal requests -- a row segment is read by this CU once and by no other
            // workgroup when the batch is a single chain group (several groups share rows through L2)
            if (F32 && NTH)
                asm("s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dword %1, %2 nt"
                    : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst));
            else if (F32)
                asm("s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dword %1, %2"
                    : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst));
            else if (NTH)
                asm("s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, %2 nt"
                    : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst));
            else
                asm("s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, %2"
                    : "+s"(tk) : "v"(voff), "s"(rowp), "s"(dst));
        };
        const int kstr = a.ustride / LW;
        const char *const cnt_base = reinterpret_cast<const char *>(a.ucount + gt * a.vmax);
        const char *const ent_base = reinterpret_cast<const char *>(a.uent + ((gt * a.vmax) * LW + lw) * kstr * 2);
        const uint32_t ent_step = (uint32_t)(a.ustride * 8);
        int U_a;
        uint32_t rid[KPRE], rsl[KPRE];
        const elem_t *G_a = nullptr;
        auto fetch_ids = [&](int p, int iv) {
            // constant address space: the tables are written by k_gf_group_tables before this
            // launch, never here, and must come in through scalar loads -- a vector load would be
            // counted in this wavefront's vmcnt together with its row requests
            typedef const __attribute__((address_space(4))) uint32_t *cu32;
            typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
            typedef const __attribute__((address_space(4))) u32x16 *cent;
            U_a = (int)*(cu32)(uintptr_t)(cnt_base + (uint32_t)p * 4u);
            const char *e = ent_base + (uint32_t)p * ent_step;
            const u32x16 e0 = *(cent)(uintptr_t)e;
            const u32x16 e1 = *(cent)(uintptr_t)(e + 64);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                rid[k] = e0[2 * k];      rsl[k] = e0[2 * k + 1];
                rid[8 + k] = e1[2 * k];  rsl[8 + k] = e1[2 * k + 1];
            }
            if constexpr (F32 != 0) G_a = a.G32[iv] + tbase; else G_a = a.G[iv] + tbase;
        };
        auto dma_count = [&](int U) { return U > lw ? (U - lw + LW - 1) / LW : 0; };
        // rows lw, lw + LW, ... of the list whose entries are in rid / rsl -> buffer at boff
        auto issue_rows = [&](int p, int boff) {
            if (dma_lane) {
                uint32_t tk = 0;
#pragma unroll
                for (int k = 0; k < KPRE; k++)
                    if (lw + k * LW < U_a) dma_row(G_a, rid[k], rsl[k], boff, tk);
                if (U_a > KPRE * LW) {   // rare: more than 64 distinct rows
                    typedef const __attribute__((address_space(4))) uint32_t *cu32;
                    cu32 ue = (cu32)(uintptr_t)(ent_base + (uint32_t)p * ent_step);
                    for (int k = KPRE; lw + k * LW < U_a; k++) dma_row(G_a, ue[2 * k], ue[2 * k + 1], boff, tk);
                }
                keep |= tk;
            }
        };
        // steps 0 .. NB-2 go out before the loop, then step s+NB-1 behind the barrier of step s
        int boff[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) boff[j] = j * bufsz;   // boff[j]: buffer of step s+j
        int cnt[NB];                                          // cnt[j]: this loader's requests of step s+j
#pragma unroll
        for (int j = 0; j < NB; j++) cnt[j] = 0;
        int pn = 0, ivn = 0;                                  // next step to fetch ids for
        fetch_ids(pn, ivn);
#pragma unroll
        for (int j = 0; j < NB - 1; j++) {
            if (j < nsteps) {
                issue_rows(pn, boff[j]);
                cnt[j] = dma_count(U_a);
            }
            advance(pn, ivn);
            fetch_ids(pn, ivn);                               // ids of step j+1
        }
        int p_i = pn;                                         // step whose ids are in rid / rsl (s+NB-1)
        for (int s = 0; s < nsteps; s++) {
            __builtin_amdgcn_sched_barrier(0);
            // this loader's rows of step s have landed when at most its requests of the NB-2
            // younger steps stay in flight (s_waitcnt takes an immediate; an over-wait is safe)
            {
                int young = 0;
#pragma unroll
                for (int j = 1; j < NB - 1; j++) young += cnt[j];
                uint32_t tk = 0;
#define BA_LWAIT(NOUT) asm("s_waitcnt vmcnt(" #NOUT ")" : "+s"(tk) : "s"(s))
                switch (young) {
                case 0: BA_LWAIT(0); break;   case 1: BA_LWAIT(1); break;   case 2: BA_LWAIT(2); break;
                case 3: BA_LWAIT(3); break;   case 4: BA_LWAIT(4); break;   case 5: BA_LWAIT(5); break;
                case 6: BA_LWAIT(6); break;   case 7: BA_LWAIT(7); break;   case 8: BA_LWAIT(8); break;
                case 9: BA_LWAIT(9); break;   case 10: BA_LWAIT(10); break; case 11: BA_LWAIT(11); break;
                case 12: BA_LWAIT(12); break; case 13: BA_LWAIT(13); break; case 14: BA_LWAIT(14); break;
                case 15: BA_LWAIT(15); break; case 16: BA_LWAIT(16); break; case 17: BA_LWAIT(17); break;
                case 18: BA_LWAIT(18); break; case 19: BA_LWAIT(19); break; case 20: BA_LWAIT(20); break;
                case 21: BA_LWAIT(21); break; case 22: BA_LWAIT(22); break; case 23: BA_LWAIT(23); break;
                case 24: BA_LWAIT(24); break; case 25: BA_LWAIT(25); break; case 26: BA_LWAIT(26); break;
                case 27: BA_LWAIT(27); break; case 28: BA_LWAIT(28); break; case 29: BA_LWAIT(29); break;
                case 30: BA_LWAIT(30); break; case 31: BA_LWAIT(31); break; default: BA_LWAIT(32); break;
                }
#undef BA_LWAIT
                keep |= tk;
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();   // rows of step s published; buffer of step s-1 is free
            __builtin_amdgcn_sched_barrier(0);
            int k_new = 0;
            if (s + NB - 1 < nsteps) {
                issue_rows(p_i, boff[NB - 1]);
                k_new = dma_count(U_a);
            }
            advance(pn, ivn);
            p_i = pn;
            fetch_ids(pn, ivn);
            const int b0 = boff[0];
#pragma unroll
            for (int j = 0; j < NB - 1; j++) { boff[j] = boff[j + 1]; cnt[j] = cnt[j + 1]; }
            boff[NB - 1] = b0;
            cnt[NB - 1] = 0;
            cnt[NB - 2] = k_new;
        }
        __builtin_amdgcn_s_barrier();   // the consumers' barrier "of step nsteps" (their pipelined loop)
        // the consumers' epilogue passes the data tile through LDS behind two barriers
        if (MODE != GF_STORE_SYN) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        if (keep != 0) a.out[0] = 0.0;   // never true: keeps the request statements alive
        return;
    }

    // ==================================== consumer ====================================
    const int64_t c = a.order ? (int64_t)a.order[g * CG + tid] : g * CG + tid;
    // the zero row of every buffer (slot index ucap): see k_gfstack_ws
    for (int i = tid; i < NB * GS_PITCH; i += CG)
        reinterpret_cast<elem_t *>(xbuf)[(i / GS_PITCH) * bufsz + a.ucap * GS_PITCH + (i % GS_PITCH)] = (elem_t)0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    double acc[GS_NT];
#pragma unroll
    for (int i = 0; i < GS_NT; i++) acc[i] = 0.0;
    // the lane's slot and weight, fetched one step ahead by asm loads hipcc does not count
    uint32_t sl_n;
    double wl_n;
    const uint32_t voff_s = (uint32_t)tid * 2u;
    const char *const slot_base = reinterpret_cast<const char *>(a.slot + (gt * a.vmax) * CG);
    const char *const w_base = reinterpret_cast<const char *>(a.w + (gt * a.vmax) * CG);
    const uint32_t slot_step = (uint32_t)(CG * 2);
    const uint32_t w_step = (uint32_t)(CG * 8);
    const int64_t w_var_bytes = a.w_var_stride * 8;
    auto tab_slot = [&](int p) { return slot_base + (uint32_t)p * slot_step; };
    auto tab_w = [&](int p, int iv) {
        return w_base + (uint32_t)p * w_step + (nvar == 1 ? (int64_t)0 : (int64_t)iv * w_var_bytes);
    };
    auto fetch_tabs = [&](const char *ps, const char *pw) {
        asm("global_load_ushort %0, %1, %2" : "=v"(sl_n) : "v"(voff_s), "s"(ps));
        uint32_t voff_w;
        asm("v_lshlrev_b32 %1, 2, %2\n\t"
            "global_load_dwordx2 %0, %1, %3" : "=v"(wl_n), "=&v"(voff_w) : "v"(voff_s), "s"(pw));
    };
    // xs = LDS byte address of the lane's row = base + slot * 33 pairs (8 or 16 bytes each)
    auto row_address = [&](uint32_t base) {
        uint32_t x;
        asm("s_waitcnt vmcnt(0)\n\t"
            "v_lshl_add_u32 %0, %1, 5, %1\n\t"
            "v_lshl_add_u32 %0, %0, %c4, %2" : "=&v"(x) : "v"(sl_n), "s"(base), "v"(wl_n), "n"(F32 ? 3 : 4));
        return x;
    };
    auto landed_weight = [&](uint32_t after) {   // `after`: the row address, i.e. behind the wait
        double x;
        asm("v_mov_b64 %0, %1" : "=v"(x) : "v"(wl_n), "v"(after));
        return x;
    };
    static_assert(GS_PITCH == 66, "row_address multiplies by 33 qwords");
    static_assert(GS_NT == 64, "the read schedule below is written for 64-sample tiles");
    // The gather of a step is 8 groups of 4 ds_read_b64 (two floats each) + 8 (widen + FMA), two groups
    // in flight: HALF the LDS instructions of the f64 kernel per step -- its consumers are bound by the
    // LDS instruction rate (1081 of ~1300 cycles per step, profiles/archive/r2_variants.md), not by LDS bytes.
    // Pipelined across the step boundary like k_gfstack_ws; the FMAs are plain fma() between wait
    // statements that name the landing registers.
    pair_t ya[4], yb[4];
    int p1 = 0, iv1 = 0;          // position of the step whose slot/weight are fetched next
    advance(p1, iv1);
    int gbuf = 0;                 // row buffer of the step being gathered
    const int ring = NB * bufsz;
    double w;
    uint32_t xs;
    fetch_tabs(tab_slot(0), tab_w(0, 0));
    xs = row_address(lds0);
    w = landed_weight(xs);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();     // rows of step 0 visible
    __builtin_amdgcn_sched_barrier(0);
    lds_rd4_pair<F32 ? 1 : 2, 0>(ya, xs, -1);
    lds_rd4_pair<F32 ? 1 : 2, 32>(yb, xs, -1);
    __builtin_amdgcn_sched_barrier(0);
    fetch_tabs(tab_slot(p1), tab_w(p1, iv1));
    advance(p1, iv1);
#define F2_FMA(G, Y)                                                          \
    _Pragma("unroll") for (int q = 0; q < 4; q++) {                           \
        acc[(G) * 8 + 2 * q] = fma((double)Y[q].x, w, acc[(G) * 8 + 2 * q]);   \
        acc[(G) * 8 + 2 * q + 1] = fma((double)Y[q].y, w, acc[(G) * 8 + 2 * q + 1]); \
    }
#define F2_GROUP(G, Y, OFFNEXT)              \
    lds_wait4_pair<4>(Y);                      \
    __builtin_amdgcn_sched_barrier(0);       \
    F2_FMA(G, Y)                             \
    __builtin_amdgcn_sched_barrier(0);       \
    lds_rd4_pair<F32 ? 1 : 2, OFFNEXT>(Y, xs, s);           \
    __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(disable)
    for (int s = 0; s < nsteps; s++) {
        int gnext = gbuf + bufsz;
        if (gnext == ring) gnext = 0;
        const char *const ps2 = tab_slot(p1), *const pw2 = tab_w(p1, iv1);   // tables of step s+2
        advance(p1, iv1);
        __builtin_amdgcn_sched_barrier(0);
        F2_GROUP(0, ya, 64)
        F2_GROUP(1, yb, 96)
        F2_GROUP(2, ya, 128)
        F2_GROUP(3, yb, 160)
        F2_GROUP(4, ya, 192)
        F2_GROUP(5, yb, 224)
        // group 6: every read of this step is back once both groups in flight have landed
        lds_wait4_pair<0>(ya);
        lds_wait4_pair<0>(yb);
        __builtin_amdgcn_sched_barrier(0);
        F2_FMA(6, ya)
        __builtin_amdgcn_sched_barrier(0);
        gbuf = gnext;
        const uint32_t xs_n = row_address(lds0 + (uint32_t)(gbuf * ES));   // slot / weight of step s+1 have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // rows of step s+1 visible; this wavefront is done with buffer s
        __builtin_amdgcn_sched_barrier(0);
        lds_rd4_pair<F32 ? 1 : 2, 0>(ya, xs_n, s);     // first group of step s+1 (after the last step: unused rows)
        __builtin_amdgcn_sched_barrier(0);
        // group 7 with the weight of step s, then the weight of step s+1 and that step's second group
        F2_FMA(7, yb)
        __builtin_amdgcn_sched_barrier(0);
        w = landed_weight(xs_n);
        lds_rd4_pair<F32 ? 1 : 2, 32>(yb, xs_n, s);
        xs = xs_n;
        __builtin_amdgcn_sched_barrier(0);
        fetch_tabs(ps2, pw2);           // slot / weight of step s+2
    }
#undef F2_GROUP
#undef F2_FMA
    // drain the reads issued for the step after the last (their registers stay reserved until here)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]), "+v"(ya[3]), "+v"(yb[0]), "+v"(yb[1]), "+v"(yb[2]),
                   "+v"(yb[3]), "+v"(sl_n), "+v"(wl_n));
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: lane = chain c, acc[i] = synthetics[c, t, n0 + i]
    const bool live = (c < a.C);
    const int nvalid = (int)min((int64_t)GS_NT, N - n0);
    if (MODE == GF_STORE_SYN) {
        if (live) {
            double *o = a.out + (c * a.T + t) * N + n0;
#pragma unroll
            for (int i = 0; i < GS_NT; i++)
                if (i < nvalid) o[i] = acc[i];
        }
        return;
    }
    __builtin_amdgcn_s_barrier();
    if (tid < GS_NT) xbuf[tid] = (tid < nvalid) ? a.data[t * N + n0 + tid] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (MODE == GF_RESID_STORE) {
        double *o = a.out + (c * a.T + t) * N + n0;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 8) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i++)
                if (live && i < nvalid) o[i] = xbuf[i] - acc[i];  // seismic.py:1332
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const double w = a.wscalar[t];
        double q = 0.0;
#pragma unroll
        for (int i0 = 0; i0 < GS_NT; i0 += 8) {
#pragma unroll
            for (int i = i0; i < i0 + 8; i++)
                if (i < nvalid) {
                    const double tt = w * (xbuf[i] - acc[i]);
                    q = fma(tt, tt, q);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (live) a.partial[(c * a.T + t) * a.ntile + tile] = q;
    }
}


template <int WAVES, int NROW, int MODE>
static int launch_shared_one(dim3 grid, size_t lds, hipStream_t s, const GsArgs &a)
{
    void (*kern)(GsArgs) = nullptr;
    if constexpr (WAVES == 16) {
        kern = k_gfstack_dma<WAVES, NROW, MODE, 32, 1>;   // 1024-chain groups exist as NT = 32 only
    } else if (a.f32pair) {
        kern = k_gfstack_dmaf<WAVES, NROW, MODE>;
    } else {
        kern = (a.dma == 2 && a.nt == 32) ? k_gfstack_dma<WAVES, NROW, MODE, 32, 1>
             : (a.dma == 2) ? k_gfstack_dma<WAVES, NROW, MODE, 64, 1>
                            : k_gfstack_dma<WAVES, NROW, MODE, 64, 0>;
    }
    if (lds > 64 * 1024)
        BA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, a);
    return BEATAMD_OK;
}

template <int WAVES, int NROW>
static int launch_shared_mode(int mode, dim3 grid, size_t lds, hipStream_t s, const GsArgs &a)
{
    if (mode == GF_STORE_SYN) return launch_shared_one<WAVES, NROW, GF_STORE_SYN>(grid, lds, s, a);
    if (mode == GF_RESID_SCALAR) return launch_shared_one<WAVES, NROW, GF_RESID_SCALAR>(grid, lds, s, a);
    return launch_shared_one<WAVES, NROW, GF_RESID_STORE>(grid, lds, s, a);
}

template <int WAVES>
static int launch_shared_nrow(int nrow, int mode, dim3 grid, size_t lds, hipStream_t s,
                              const GsArgs &a)
{
    if (nrow == 1) return launch_shared_mode<WAVES, 1>(mode, grid, lds, s, a);
    return launch_shared_mode<WAVES, 4>(mode, grid, lds, s, a);
}

// (single-row interpolation only: with four rows per chain the consumers run out of registers and
// hipcc spills the not-yet-landed results of the hidden table loads)
// pair: 0 k_gfstack_ws, 1 the float-storage pair gather, 2 the float64 pair gather (A/B)
static int launch_ws(int mode, dim3 grid, size_t lds, hipStream_t s, const GsArgs &a, int pair)
{
    void (*kern)(GsArgs);
#define BA_WS_PICK(NAME, ...)                                                                          \
    (mode == GF_STORE_SYN ? NAME<__VA_ARGS__ GF_STORE_SYN, 3, 1>                                      \
     : mode == GF_RESID_SCALAR ? NAME<__VA_ARGS__ GF_RESID_SCALAR, 3, 1> : NAME<__VA_ARGS__ GF_RESID_STORE, 3, 1>)
#define BA_WS_PICK0(NAME, ...)                                                                         \
    (mode == GF_STORE_SYN ? NAME<__VA_ARGS__ GF_STORE_SYN, 3, 0>                                      \
     : mode == GF_RESID_SCALAR ? NAME<__VA_ARGS__ GF_RESID_SCALAR, 3, 0> : NAME<__VA_ARGS__ GF_RESID_STORE, 3, 0>)
    if (pair == 1) kern = a.nthint ? BA_WS_PICK(k_gfstack_wsp, 1,) : BA_WS_PICK0(k_gfstack_wsp, 1,);
    else if (pair == 2) kern = a.nthint ? BA_WS_PICK(k_gfstack_wsp, 0,) : BA_WS_PICK0(k_gfstack_wsp, 0,);
    else if (mode == GF_RESID_BAND1) kern = a.nthint ? k_gfstack_ws<1, GF_RESID_BAND1, 3, 1> : k_gfstack_ws<1, GF_RESID_BAND1, 3, 0>;
    else kern = a.nthint ? BA_WS_PICK(k_gfstack_ws, 1,) : BA_WS_PICK0(k_gfstack_ws, 1,);
#undef BA_WS_PICK0
#undef BA_WS_PICK
    BA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(768), lds, s, a);
    return BEATAMD_OK;
}

// ---------------------------------------------------------------------------------------------
// Tables of k_gfstack_ws / k_gfstack_wsp (round 5): ROW PASSES.  The kernel walks a list of "vsteps"; a vstep
// stages at most `cap` (32, 64 or 96) distinct rows in one LDS row buffer.  A (chain group, target, patch) whose
// chains touch U <= cap distinct rows is one vstep, as before; one that touches more -- a library on a fine
// (duration x start-time) grid: the reference's tutorial grid gives 178 rows per patch for 512 prior chains -- is
// cut into ceil(U / cap) PASSES of equal size over the rows in ascending order.  A chain takes part in the pass
// that holds its row; in the other passes of the patch its lane reads the buffer's ZERO ROW (slot index `cap`,
// written once by the kernel) with weight 0: fma(0, 0, acc) = acc exactly, and since every chain still sees its
// rows in ascending patch order the kernel stays bitwise equal to k_gfstack.  Nothing is sized by the library's
// D * S any more: the distinct rows are found by ranking the 512 row ids of the group against each other.
//   k_ws_tables<0>  per (group, target, patch): U -> number of passes            (only when passes can occur)
//   k_ws_scan       per (group, target): first vstep of every patch, total
//   k_ws_tables<1>  per (group, target, patch): row lists, LDS slots (by bank window, as k_gf_group_tables),
//                   per-lane slot and weight of every pass
constexpr int WS_CG = 512;        // chains per group (8 consumer wavefronts)
constexpr int WS_LW = 4;          // loader wavefronts
constexpr int WS_USTRIDE = 128;   // list entries per vstep in the row-request table (cap <= 96)
constexpr int WS_CAP_MAX = 96;    // three buffers of 96 + 1 slots x 520 B = 151 KB of the CU's 160 KB
constexpr int WS_MAXPASS = (WS_CG + WS_CAP_MAX - 1) / WS_CAP_MAX;
static bool ws_wanted(const GfStackCall &k, int CG);

struct WsTabArgs {
    int nvar, cap;
    int64_t C, T, P, DS;      // T: targets the tables are built for (1 or all); DS: library rows per (target, patch)
    int64_t vmax;             // vsteps per (group, target) the tables are strided by
    const uint32_t *rowoff;   // [C,T,P] global row ids (k_gf_tables)
    ChainVec slips[4];
    const uint32_t *order;    // [ngroups*512] chain at position i of the batch, ~0 behind the last (nullptr: i)
    uint32_t *npass;          // [gtp] passes of the patch (phase 0 out)
    const uint32_t *voff;     // [gtp] first vstep of the patch; nullptr: one pass per patch, vstep = patch
    uint32_t *utotal;         // [gtp] distinct rows (statistics)
    uint32_t *ucount;         // [gt*vmax + v] rows of the vstep
    uint32_t *uent;           // [gt*vmax + v][loader][WS_USTRIDE / WS_LW][2] = (row id, LDS slot)
    uint16_t *slot;           // [gt*vmax + v][WS_CG]
    double *w;                // [variable][gt*vmax + v][WS_CG]
    int64_t w_var_stride;
    int64_t R;                // patch split: slot t covers patches (t % R) * P + p of the real model (slips)
};

__global__ void __launch_bounds__(256) k_ws_scan(const uint32_t *npass, uint32_t *voff, uint32_t *nv, int64_t P)
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
    if (tid == 0) nv[gt] = run;
}

// MAP: the distinct rows through a presence map over the D * S rows of the patch in LDS (libraries up to WS_MAP_MAX rows
// per patch: every one so far); else by ranking the 512 row ids against each other (no bound, ~20 x the LDS traffic)
constexpr int64_t WS_MAP_MAX = 16384;
template <int FILL, int MAP>
__global__ void __launch_bounds__(WS_CG) k_ws_tables(WsTabArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t rowmap[];   // MAP: [DS] presence -> position
    __shared__ __attribute__((aligned(16))) uint32_t vals[WS_CG];   // the lanes' row ids (dead lanes: ~0)
    __shared__ __attribute__((aligned(16))) uint32_t firstv[WS_CG]; // the row id where the lane is the first to name it, else ~0
    __shared__ uint32_t lst[WS_CG];     // distinct rows, ascending
    __shared__ uint32_t gm[WS_CG];      // 32-lane groups of the workgroup that use distinct row i
    __shared__ uint16_t slt[WS_CG];     // LDS slot of distinct row i inside its pass
    __shared__ uint8_t rk2i[WS_MAXPASS][128];   // per pass: popularity rank -> row of the pass
    __shared__ uint32_t wsum[WS_CG / 64 + 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int64_t gtp = xcd_items8(blockIdx.x, gridDim.x);       // (g*T + t)*P + p
    const int64_t p = gtp % a.P;
    const int64_t gt = gtp / a.P;
    const int64_t t = gt % a.T;
    const int64_t g = gt / a.T;
    const int64_t pos_in_batch = g * WS_CG + tid;
    const bool live = pos_in_batch < a.C;
    const int64_t c = live ? (a.order ? (int64_t)a.order[pos_in_batch] : pos_in_batch) : 0;
    const uint32_t v = live ? a.rowoff[(c * a.T + t) * a.P + p] : 0xffffffffu;
    // (the chain's slips are requested here, next to its row id: behind the barriers below they were one more exposed round
    // trip per patch)
    double sl[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (FILL)
        for (int iv = 0; iv < a.nvar; iv++)
            if (live) sl[iv] = a.slips[iv].base[c * a.slips[iv].stride + a.slips[iv].off + (t % a.R) * a.P + p];
    gm[tid] = 0;
    uint32_t pos = 0, U = 0;
    bool first = false;
    if constexpr (MAP) {
        const uint32_t row0 = (uint32_t)((t * a.P + p) * a.DS);
        for (int64_t i = tid; i < a.DS; i += WS_CG) rowmap[i] = 0;
        __syncthreads();
        if (live) rowmap[v - row0] = 1;   // benign race: every writer stores 1
        __syncthreads();
        // exclusive scan of the map in row order: chunks of 512 flags, rank inside a wavefront by ballot / popcount
        uint32_t run = 0;
        for (int64_t base = 0; base < a.DS; base += WS_CG) {
            const int64_t i = base + tid;
            const uint32_t f = (i < a.DS) ? rowmap[i] : 0u;
            const uint64_t m = __ballot(f != 0);
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int q = 0; q < WS_CG / 64; q++) {
                const uint32_t x = wsum[q];
                if (q < wv) before += x;
                total += x;
            }
            if (f) {
                const uint32_t ps = run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                rowmap[i] = (uint16_t)ps;
                if constexpr (FILL) lst[ps] = row0 + (uint32_t)i;
            }
            run += total;
            __syncthreads();
        }
        U = run;
        if (live) pos = rowmap[v - row0];
    } else {
        vals[tid] = v;
        __syncthreads();
        // first lane to name its row
        first = live;
        {
            const uint4 *v4 = reinterpret_cast<const uint4 *>(vals);
            for (int k4 = 0; k4 < WS_CG / 4; k4++) {
                const uint4 x = v4[k4];
                const int k = 4 * k4;
                first = first && !((x.x == v && k < tid) || (x.y == v && k + 1 < tid) || (x.z == v && k + 2 < tid) ||
                                   (x.w == v && k + 3 < tid));
            }
        }
        firstv[tid] = first ? v : 0xffffffffu;
        {
            const uint64_t m = __ballot(first);
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        // position of the lane's row among the distinct rows in ascending order
        if (live) {
            const uint4 *f4 = reinterpret_cast<const uint4 *>(firstv);
            for (int k4 = 0; k4 < WS_CG / 4; k4++) {
                const uint4 x = f4[k4];
                pos += (x.x < v) + (x.y < v) + (x.z < v) + (x.w < v);
            }
        }
        for (int q = 0; q < WS_CG / 64; q++) U += wsum[q];
    }
    const int npass = (int)((U + (uint32_t)a.cap - 1) / (uint32_t)a.cap);
    if constexpr (!FILL) {
        if (tid == 0) {
            a.npass[gtp] = (uint32_t)npass;
            a.utotal[gtp] = U;
        }
        return;
    }
    if (tid == 0) a.utotal[gtp] = U;
    const int per = ((int)U + npass - 1) / npass;       // rows of a pass (the last may hold fewer)
    if constexpr (!MAP) {
        if (first) lst[pos] = v;
    }
    if (live) atomicOr(&gm[pos], 1u << (tid >> 5));
    __syncthreads();
    const int mypass = live ? (int)pos / per : -1;
    // ---- LDS slots of the rows of pass k: wavefront k.  k_gfstack_ws reads row `slot` of a lane at slot * 65 doubles
    // with ds_read_b64: the 32 lanes of a lane group hit the two-bank window (slot + sample) mod 32, so rows whose slots
    // differ by 32 collide when ONE lane group reads both.  Rows in order of decreasing number of lane groups using
    // them; the first 32 take a window each, every further row the window whose occupants share the fewest lane groups
    // with it (then the emptiest, then the lowest).  Results never depend on the slots, only the LDS read timing does.
    if (wv < npass) {
        const int base = wv * per;
        const int n = min(per, (int)U - base);
        if (n <= 32) {
            if (lane < n) slt[base + lane] = (uint16_t)lane;
        } else {
            const int depth = a.cap / 32;
            const int r0 = lane, r1 = lane + 64;
            const uint32_t m0 = r0 < n ? gm[base + r0] : 0u, m1 = r1 < n ? gm[base + r1] : 0u;
            const int key0 = r0 < n ? ((__popc(m0) << 8) | (255 - r0)) : -1;
            const int key1 = r1 < n ? ((__popc(m1) << 8) | (255 - r1)) : -1;
            int rank0 = 0, rank1 = 0;
            for (int j = 0; j < n; j++) {
                const int kj = (__popc(gm[base + j]) << 8) | (255 - j);
                rank0 += kj > key0;
                rank1 += kj > key1;
            }
            if (r0 < n && rank0 < 32) slt[base + r0] = (uint16_t)rank0;
            if (r1 < n && rank1 < 32) slt[base + r1] = (uint16_t)rank1;
            if (r0 < n) rk2i[wv][rank0] = (uint8_t)r0;
            if (r1 < n) rk2i[wv][rank1] = (uint8_t)r1;
            wave_lds_fence();
            uint32_t wmask = 0;
            int wcnt = 0;
            if (lane < 32) {
                wmask = gm[base + rk2i[wv][lane]];
                wcnt = 1;
            }
            for (int rk = 32; rk < n; rk++) {
                const int ridx = (int)rk2i[wv][rk];
                const uint32_t mm = gm[base + ridx];
                int cost = (lane < 32 && wcnt < depth) ? ((__popc(wmask & mm) << 12) | (wcnt << 6) | lane) : 0x7fffffff;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cost = min(cost, __shfl_xor(cost, off, 64));
                const int w = cost & 63;
                if (lane == w) {
                    slt[base + ridx] = (uint16_t)(w + 32 * wcnt);
                    wmask |= mm;
                    wcnt++;
                }
            }
        }
    }
    __syncthreads();
    const int64_t v0 = gt * a.vmax + (a.voff ? (int64_t)a.voff[gtp] : p);
    const uint16_t myslot = live ? slt[pos] : (uint16_t)0;
    constexpr int kstr = WS_USTRIDE / WS_LW;
    for (int k = 0; k < npass; k++) {
        const int64_t vs = v0 + k;
        const int base = k * per;
        const int n = min(per, (int)U - base);
        const bool mine = mypass == k;
        a.slot[vs * WS_CG + tid] = mine ? myslot : (uint16_t)a.cap;     // (not in this pass / no chain: the zero row)
        for (int iv = 0; iv < a.nvar; iv++) a.w[(int64_t)iv * a.w_var_stride + vs * WS_CG + tid] = mine ? sl[iv] : 0.0;
        if (tid == 0) a.ucount[vs] = (uint32_t)n;
        if (tid < WS_USTRIDE) {
            // (row id, LDS slot) of list entry j, grouped by the loader that stages it (entry j: loader j mod 4);
            // padded with the last id: the loaders fetch entries ahead of the count
            const int j = tid;
            uint32_t *e = a.uent + ((vs * WS_LW + (j % WS_LW)) * kstr + (j / WS_LW)) * 2;
            e[0] = lst[base + (j < n ? j : n - 1)];
            e[1] = j < n ? (uint32_t)slt[base + j] : 0u;
        }
    }
}

// chains per group for a batch of C chains: the group size that minimises
// (number of groups) x (cost of one group's launch share).  Relative costs measured on config 3
// (ms per single-group launch: 64 chains 2.4, 128 2.9, 256 3.6, 512 6.2): a larger group is much
// cheaper per chain (the distinct rows are staged once for more lanes), so padding a batch up to
// the next size usually beats splitting it (192 chains: one 256-group 3.6 ms, three 64-groups 6.9 ms).
static int pick_group(int64_t C)
{
    const int cand[4] = {512, 256, 128, 64};
    const double cost[4] = {2.55, 1.5, 1.2, 1.0};
    int best = 64;
    double bestc = 1e300;
    for (int i = 0; i < 4; i++) {
        const double c = (double)((C + cand[i] - 1) / cand[i]) * cost[i];
        if (c < bestc - 1e-12) { bestc = c; best = cand[i]; }
    }
    return best;
}

// the distinct-row bound of chain groups of `cg`, or false when their row buffers or the table
// kernel's maps do not fit LDS
static bool shared_fit(const GfStackCall &k, int cg, int *ucap_out)
{
    const SeisLib &L = *k.libs[0];
    const int nrow = k.interp == BEATAMD_MULTILINEAR ? 4 : 1;
    const int64_t DS = L.D * L.S;
    const int64_t ucap = std::min<int64_t>((int64_t)cg * nrow, DS);
    if (ws_wanted(k, cg)) {   // row passes: any library
        *ucap_out = (int)std::max<int64_t>(ucap, 2);
        return true;
    }
    if (2 * ucap * (GS_NT_MAX + 2) * 8 > 158 * 1024) return false;   // two row buffers of the bound
    if (2 * DS * 4 + cg * 4 + 2048 > 60 * 1024) return false;         // presence map + group masks of k_gf_group_tables
    *ucap_out = (int)std::max<int64_t>(ucap, 2);
    return true;
}

// group sizes worth timing for this call (gfstack.hip tunes the choice once per problem shape)
int gfstack_shared_candidates(const GfStackCall &k, int *cgs, int *ucaps)
{
    int n = 0;
    const int cand[4] = {512, 256, 128, 64};
    for (int i = 0; i < 4; i++) {
        // patch split (small-N libraries): the loader / consumer kernel only (the small-group kernels index their
        // weight tables by the real patch)
        if (k.patch_split > 1 && cand[i] != 512) continue;
        // a group size that would leave more than half of its lanes without a chain only pads
        if (cand[i] > 64 && (int64_t)cand[i] / 2 >= k.C) continue;
        // large batches: small groups stage every distinct row many times over and have never been the fastest
        // (config 3, 512 chains: 512 6.2, 256 2 x 3.6, 128 4 x 2.9, 64 8 x 2.4 ms); timing them on the first call of a
        // shape cost 12 full launches (1 s at 4096 chains) -- from 1024 chains on 512 / 256 only
        if (k.C >= 1024 && cand[i] < 256) continue;
        if (k.C >= 384 && cand[i] < 128) continue;
        int u = 0;
        if (shared_fit(k, cand[i], &u)) { cgs[n] = cand[i]; ucaps[n] = u; n++; }
    }
    return n;
}

bool gfstack_shared_applicable(const GfStackCall &k, int *cg_out, int *ucap_out)
{
    const SeisLib &L = *k.libs[0];
    const int nrow = k.interp == BEATAMD_MULTILINEAR ? 4 : 1;
    if (L.N % 2 != 0) return false;
    const GfKnobs &kn = *k.knobs;
    if (GfKnobs::is(kn.gf_kernel, 0)) return false;   // 0 = streaming kernel, 1 = shared kernel
    const bool forced = GfKnobs::is(kn.gf_kernel, 1);
    if (!forced && k.C < 48) return false;  // too few chains to share rows
    int cg = pick_group(k.C);
    const int gq = GfKnobs::get(kn.gs_cg, 0);
    if (gq == 64 || gq == 128 || gq == 256 || gq == 512 || gq == 1024) cg = gq;
    if (k.patch_split > 1) {      // the loader / consumer kernel or none (the streaming kernel then)
        if (!ws_wanted(k, WS_CG)) return false;
        cg = WS_CG;
    }
    const int64_t DS = L.D * L.S;
    int64_t ucap = std::min<int64_t>((int64_t)cg * nrow, DS);
    if (ws_wanted(k, cg)) {
        *cg_out = cg;
        *ucap_out = (int)std::max<int64_t>(ucap, 2);
        return true;
    }
    // two row buffers of the group's distinct-row bound must fit LDS (prefer >= 2 workgroups per CU): halve the group
    // until they do
    const int GS_PITCH = GS_NT_MAX + 2;
    while (ucap * GS_PITCH * 8 > 72 * 1024 && cg > 64) {
        cg /= 2;
        ucap = std::min<int64_t>((int64_t)cg * nrow, DS);
    }
    if (2 * ucap * GS_PITCH * 8 > 158 * 1024) return false;
    if (2 * DS * 4 + cg * 4 + 2048 > 60 * 1024) return false;  // presence map + group masks of k_gf_group_tables
    *cg_out = cg;
    *ucap_out = (int)std::max<int64_t>(ucap, 2);
    return true;
}

// 512-chain groups with one row per chain and patch: the loader / consumer kernel, whatever the library's
// (duration x start-time) grid (row passes, k_ws_tables)
static bool ws_wanted(const GfStackCall &k, int CG)
{
    const GfKnobs &kn = *k.knobs;
    const bool want = GfKnobs::get(kn.gs_ws, GS_WS_DEFAULT) != 0;
    return want && CG == WS_CG && k.interp != BEATAMD_MULTILINEAR && k.libs[0]->N % 2 == 0 &&
           !(GfKnobs::set(kn.gs_dma) && kn.gs_dma != 2) && !(GfKnobs::set(kn.gs_nt) && kn.gs_nt != 64);
}

static int launch_gfstack_ws(beatamd_ctx *ctx, const GfStackCall &k, const uint32_t *rowoff, int64_t Ttab)
{
    const GfKnobs &kn = *k.knobs;
    const SeisLib &L = *k.libs[0];
    const int64_t ngroups = (k.C + WS_CG - 1) / WS_CG;
    const int64_t GT = ngroups * Ttab, GTP = GT * L.P;
    // row slots of an LDS buffer: what a patch can touch at most, in whole bank windows, up to the 96 that three
    // buffers leave room for; a patch that touches more is staged in passes
    const int64_t bound = std::min<int64_t>(WS_CG, L.D * L.S);
    const int cap = bound <= 32 ? 32 : bound <= 64 ? 64 : WS_CAP_MAX;
    const int maxpass = (int)((bound + cap - 1) / cap);
    const int64_t vmax = L.P * maxpass;
    void *p = nullptr;

    WsTabArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.nvar = k.nvar; ta.cap = cap;
    ta.C = k.C; ta.T = Ttab; ta.P = L.P; ta.DS = L.D * L.S; ta.vmax = vmax;
    ta.rowoff = rowoff;
    for (int v = 0; v < k.nvar; v++) ta.slips[v] = k.slips[v];
    ta.R = k.patch_split;
    // several groups: cut the batch into its groups by bisection along the order keys (the fused model path hands the
    // hypocentre): a compact piece of the fault per group = fewer distinct rows to stage per group and patch.  Scheduling only.
    if (ngroups > 1 && k.order_key[0].base && k.order_key[1].base && GfKnobs::get(kn.gc_global, 1) != 0)
        BA_TRY(launch_chain_members(ctx, k.C, k.order_key, WS_CG, ngroups, &ta.order, GfKnobs::get(kn.gc_global, 1) == 2));
    // [utotal GTP][npass GTP][voff GTP][nv GT]
    BA_TRY(ctx->get_scratch(SL_GS_UCOUNT, (size_t)(3 * GTP + GT) * sizeof(uint32_t), &p));
    ta.utotal = (uint32_t *)p;
    ta.npass = ta.utotal + GTP;
    uint32_t *voff = ta.npass + GTP, *nv = voff + GTP;
    // (+ 3 vsteps of padding behind the tables: the kernel runs its table pointers past the last step)
    const size_t nvs = (size_t)GT * vmax + 3;
    BA_TRY(ctx->get_scratch(SL_GS_UROWS, nvs * sizeof(uint32_t), &p));
    ta.ucount = (uint32_t *)p;
    BA_TRY(ctx->get_scratch(SL_GS_USLOT, nvs * WS_USTRIDE * 2 * sizeof(uint32_t), &p));
    ta.uent = (uint32_t *)p;
    BA_TRY(ctx->get_scratch(SL_GS_SLOT, nvs * WS_CG * sizeof(uint16_t), &p));
    ta.slot = (uint16_t *)p;
    ta.w_var_stride = (int64_t)GT * vmax * WS_CG;
    BA_TRY(ctx->get_scratch(SL_GS_W, ((size_t)ta.w_var_stride * k.nvar + (size_t)3 * WS_CG) * sizeof(double), &p));
    ta.w = (double *)p;
    {
        ScopedTimer tm(ctx, "grouptables");
        const bool map = ta.DS <= WS_MAP_MAX && !GfKnobs::is(kn.ws_map, 0);   // (BEATAMD_WS_MAP=0: tests of the ranking path)
        const size_t mlds = map ? (size_t)ta.DS * sizeof(uint16_t) : 0;
        if (maxpass > 1) {
            if (map) hipLaunchKernelGGL((k_ws_tables<0, 1>), dim3((unsigned)GTP), dim3(WS_CG), mlds, ctx->stream, ta);
            else hipLaunchKernelGGL((k_ws_tables<0, 0>), dim3((unsigned)GTP), dim3(WS_CG), 0, ctx->stream, ta);
            hipLaunchKernelGGL(k_ws_scan, dim3((unsigned)GT), dim3(256), 0, ctx->stream, ta.npass, voff, nv, L.P);
            ta.voff = voff;
        }
        if (map) hipLaunchKernelGGL((k_ws_tables<1, 1>), dim3((unsigned)GTP), dim3(WS_CG), mlds, ctx->stream, ta);
        else hipLaunchKernelGGL((k_ws_tables<1, 0>), dim3((unsigned)GTP), dim3(WS_CG), 0, ctx->stream, ta);
    }
    BA_HIP(hipGetLastError());

    GsArgs a;
    memset(&a, 0, sizeof(a));
    bool f32 = k.f32;   // float copies: the pair gather of k_gfstack_wsp<1>
    const bool pair64 = GfKnobs::is(kn.gs_pair, 1);   // A/B: ds_read_b128 pairs
    for (int v = 0; v < k.nvar; v++) {
        a.G[v] = k.libs[v]->g;
        a.G32[v] = k.libs[v]->g32;
        f32 = f32 && a.G32[v] != nullptr;
    }
    a.nvar = k.nvar; a.nrow = 1;
    a.C = k.C; a.T = L.T; a.P = L.P; a.N = L.N;
    a.Ttab = Ttab; a.rows_per_target = L.P * L.D * L.S;
    a.tslot = k.tslot;
    a.CG = WS_CG; a.ucap = cap; a.ustride = WS_USTRIDE;
    a.nt = 64; a.dma = 2; a.ws = 3;
    a.ngroups = ngroups;
    a.ntile = (int)((L.N + 63) / 64);
    a.nv = maxpass > 1 ? nv : nullptr;
    a.vmax = vmax;
    a.order = ta.order;
    a.uent = ta.uent; a.ucount = ta.ucount; a.slot = ta.slot; a.w = ta.w;
    a.w_var_stride = ta.w_var_stride;
    a.data = k.data; a.wscalar = k.wscalar; a.out = k.out;
    if (k.mode == GF_RESID_SCALAR || k.mode == GF_RESID_BAND1) {
        BA_TRY(ctx->get_scratch(SL_PARTIAL, (size_t)k.C * L.T * a.ntile * sizeof(double), &p));
        a.partial = (double *)p;
    }
    if (k.mode == GF_RESID_BAND1) {
        BA_TRY(ctx->get_scratch(SL_EDGES, (size_t)k.C * L.T * a.ntile * 2 * sizeof(double), &p));
        a.edges = (double *)p;
        a.band_w = k.band_w;
    }
    a.nthint = GfKnobs::set(kn.gs_nthint) ? (kn.gs_nthint != 0) : (GS_NTHINT_DEFAULT && ngroups == 1);
    int64_t nblocks = ngroups * L.T * a.ntile;
    {
        // chain groups of one (target, tile) on one XCD (several groups only)
        a.xcd_order = (ngroups > 1 && !GfKnobs::is(kn.gs_order, 0)) ? 1 : 0;
        if (GfKnobs::is(kn.gs_order, 1)) a.xcd_order = 1;
        if (a.xcd_order) nblocks = ((L.T * a.ntile + 7) / 8) * 8 * ngroups;
    }
    BA_CHECK(nblocks < (int64_t)0x7fffffff, BEATAMD_EINVAL, "gfstack: batch too large");
    // three row buffers of cap slots + the zero row each
    size_t lds = f32 ? (size_t)(cap + 1) * (64 + 2) * sizeof(float) * 3
                     : (size_t)(cap + 1) * (64 + (pair64 ? 2 : 1)) * sizeof(double) * 3;
    lds = std::max<size_t>(lds, (64 + 8 * 64 * 17) * sizeof(double));   // the epilogue's data tile + the residual-store tiles of the 8 consumers
    BA_CHECK(lds <= 160 * 1024, BEATAMD_EINVAL, "internal: k_gfstack_ws row buffers exceed LDS");
    if (f32 || pair64)
        snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "k_gfstack_ws%s<%d,%d,%d>", f32 ? "32" : "p64", k.mode, a.ws,
                 a.nthint);
    else
        snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "k_gfstack_ws<%d,%d,%d,%d>", 1, k.mode, a.ws, a.nthint);
    snprintf(ctx->gf_plan, sizeof(ctx->gf_plan),
             "loader/consumer kernel: 512-chain groups, nearest neighbour; %d row slots per LDS buffer (a patch can touch min(512, "
             "D*S = %lld) rows), %s", cap, (long long)(L.D * L.S),
             maxpass > 1 ? "patches that touch more are staged in passes of equal size" : "one pass per patch");
    ctx->gs_ngtp = GTP;
    ctx->gs_trep = (double)L.T / (double)Ttab;
    ctx->gs_N = L.N;
    ctx->gs_cg = WS_CG;
    ctx->gs_nvar = k.nvar;
    ctx->gs_has_passes = maxpass > 1;
    {
        ScopedTimer tm(ctx, "gfstack");
        BA_TRY(launch_ws(k.mode, dim3((unsigned)nblocks), lds, ctx->stream, a, f32 ? 1 : (pair64 ? 2 : 0)));
    }
    BA_HIP(hipGetLastError());
    if (k.mode == GF_RESID_SCALAR) BA_TRY(launch_sum_tiles(ctx, a.partial, k.C * L.T, a.ntile, k.quad));
    if (k.mode == GF_RESID_BAND1)
        BA_TRY(launch_sum_tiles_band1(ctx, a.partial, a.edges, k.band_w, k.C, L.T, L.N, a.ntile, 64, k.quad));
    return BEATAMD_OK;
}

int launch_gfstack_shared(beatamd_ctx *ctx, const GfStackCall &k_in, const uint32_t *rowoff,
                          const double *fac, int CG, int ucap, int64_t Ttab)
{
    GfStackCall k = k_in;
    if (k.mode == GF_RESID_BAND1) {
        // the bidiagonal epilogue exists in k_gfstack_ws (float64 rows); every other kernel stores the residuals and
        // launch_gfstack runs k_quadform_banded behind it (BEATAMD_QF_FUSE=0: always -- A/B, tests)
        const GfKnobs &kn0 = *k.knobs;
        bool f32 = k.f32;
        for (int v = 0; v < k.nvar; v++) f32 = f32 && k.libs[v]->g32 != nullptr;
        const bool fuse = ws_wanted(k, CG) && !f32 && !GfKnobs::is(kn0.gs_pair, 1) && !GfKnobs::is(kn0.qf_fuse, 0);
        if (!fuse) k.mode = GF_RESID_STORE;
        ctx->gf_band_fused = fuse;
    }
    if (ws_wanted(k, CG)) return launch_gfstack_ws(ctx, k, rowoff, Ttab);
    const GfKnobs &kn = *k.knobs;
    const SeisLib &L = *k.libs[0];
    const int nrow = k.interp == BEATAMD_MULTILINEAR ? 4 : 1;
    const int64_t ngroups = (k.C + CG - 1) / CG;
    const int64_t GTP = ngroups * Ttab * L.P;
    void *p = nullptr;

    GroupTabArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.nrow = nrow; ga.nvar = k.nvar; ga.CG = CG;
    ga.C = k.C; ga.T = Ttab; ga.P = L.P; ga.DS = L.D * L.S;
    ga.rowoff = rowoff; ga.fac = fac;
    for (int v = 0; v < k.nvar; v++) ga.slips[v] = k.slips[v];
    ga.ucap = ucap;
    ga.ustride = (ucap + 63) / 64 * 64;   // covers the unclamped first-pass ids of 8 waves
    BA_TRY(ctx->get_scratch(SL_GS_UROWS, (size_t)GTP * ga.ustride * sizeof(uint32_t), &p));
    ga.urows = (uint32_t *)p;
    BA_TRY(ctx->get_scratch(SL_GS_USLOT, (size_t)GTP * ga.ustride * 2 * sizeof(uint32_t), &p));
    ga.uent = (uint32_t *)p;
    BA_TRY(ctx->get_scratch(SL_GS_UCOUNT, (size_t)GTP * sizeof(uint32_t), &p));
    ga.ucount = (uint32_t *)p;
    // (+ 3 steps of padding behind both tables: k_gfstack_ws runs its table pointers past the last step)
    BA_TRY(ctx->get_scratch(SL_GS_SLOT, (size_t)(GTP + 3) * nrow * CG * sizeof(uint16_t), &p));
    ga.slot = (uint16_t *)p;
    ga.w_var_stride = (nrow == 1) ? ngroups * L.P * CG : GTP * 4 * CG;
    BA_TRY(ctx->get_scratch(SL_GS_W, ((size_t)ga.w_var_stride * k.nvar + (size_t)3 * 4 * CG) * sizeof(double), &p));
    ga.w = (double *)p;
    const bool fit_lds = CG <= 128 && !GfKnobs::is(kn.gs_fit, 0);
    ga.umax = nullptr;
    if (fit_lds) {
        BA_TRY(ctx->get_scratch(SL_GS_UMAX, sizeof(uint32_t), &p));
        ga.umax = (uint32_t *)p;
        BA_HIP(hipMemsetAsync(ga.umax, 0, sizeof(uint32_t), ctx->stream));
    }
    ga.nissue = CG / 64;
    // LDS slots by bank window for the ds_read_b64 LDS-DMA kernel with large chain groups (the
    // small groups size their LDS by the measured row count and keep dense slots): the row
    // buffers then hold 32 * depth slots
    {
        const int depth = (ucap + 31) / 32;
        const bool dma2 = !(GfKnobs::is(kn.gs_dma, 0) || GfKnobs::is(kn.gs_dma, 1)) && L.N % 2 == 0;
        // (8- and 16-wavefront workgroups are alone on their CU anyway; smaller groups would lose
        // a resident workgroup to the larger row buffers -- measured: 256 chains 6.6 -> 11.2 ms)
        // (multilinear: every chain reads four rows per patch, the lane-group masks are dense and
        // the windows bring nothing: measured 21.1 vs 20.2 ms)
        ga.windowed = (CG >= 512 && nrow == 1 && dma2 && ucap > 32 && ucap <= 128 && !GfKnobs::is(kn.gs_win, 0) &&
                       (size_t)2 * 32 * depth * (GS_NT_MAX + 2) * 8 <= 158 * 1024) ? 1 : 0;
        ga.depth = depth;
        if (ga.windowed) ucap = 32 * depth;
    }
    {
        ScopedTimer tm(ctx, "grouptables");
        const size_t lds = (size_t)(2 * ga.DS + CG + 1 + 256) * sizeof(uint32_t);
        hipLaunchKernelGGL(k_gf_group_tables, dim3((unsigned)GTP), dim3(CG), lds, ctx->stream, ga);
    }
    BA_HIP(hipGetLastError());

    GsArgs a;
    memset(&a, 0, sizeof(a));
    bool f32 = k.f32;   // float copies: k_gfstack_dmaf (LDS-DMA kernel, 64-sample tiles, groups up to 512 chains)
    for (int v = 0; v < k.nvar; v++) {
        a.G[v] = k.libs[v]->g;
        a.G32[v] = k.libs[v]->g32;
        f32 = f32 && a.G32[v] != nullptr;
    }
    a.nvar = k.nvar; a.nrow = nrow;
    a.C = k.C; a.T = L.T; a.P = L.P; a.N = L.N;
    a.Ttab = Ttab; a.rows_per_target = L.P * L.D * L.S;
    a.tslot = k.tslot;
    a.CG = CG; a.ucap = ucap; a.ustride = ga.ustride;
    a.nt = 64;
    {
        if (GfKnobs::is(kn.gs_nt, 32)) a.nt = 32;
        if (CG == 1024) a.nt = 32;
    }
    a.ngroups = ngroups;
    a.ntile = (int)((L.N + a.nt - 1) / a.nt);
    a.urows = ga.urows; a.uent = ga.uent; a.ucount = ga.ucount; a.slot = ga.slot; a.w = ga.w;
    a.w_var_stride = ga.w_var_stride;
    a.data = k.data; a.wscalar = k.wscalar; a.out = k.out;
    if (k.mode == GF_RESID_SCALAR) {
        BA_TRY(ctx->get_scratch(SL_PARTIAL, (size_t)k.C * L.T * a.ntile * sizeof(double), &p));
        a.partial = (double *)p;
    }
    int64_t nblocks = ngroups * L.T * a.ntile;
    BA_CHECK(nblocks < (int64_t)0x7fffffff, BEATAMD_EINVAL, "gfstack: batch too large");
    // Small chain groups (1-2 wavefronts per workgroup) reach the 2 waves/SIMD the kernels are
    // built for only if several workgroups fit a CU's LDS: their row buffers are sized by the
    // largest distinct-row count that actually occurs instead of the bound min(chains*rows, D*S).
    // No host synchronisation: the count of THIS batch stays on the device (ga.umax); the launcher
    // sizes the buffers from the count of the PREVIOUS launch (asynchronous read-back into a pinned
    // mailbox, used when it has arrived) plus a margin, launches that kernel guarded by "count <=
    // slots" and the full-size kernel guarded by the opposite -- exactly one of the twins works.
    int ucap_fit = 0;
    if (fit_lds) {
        if (!ctx->h_umax) {
            BA_HIP(hipHostMalloc((void **)&ctx->h_umax, sizeof(uint32_t), hipHostMallocDefault));
            BA_HIP(hipEventCreateWithFlags(&ctx->umax_event, hipEventDisableTiming));
        }
        if (ctx->umax_pending && hipEventQuery(ctx->umax_event) == hipSuccess) {
            ctx->umax_hist = (int)*ctx->h_umax;
            ctx->umax_pending = false;
        }
        (void)hipGetLastError();   // (hipErrorNotReady of the query is not an error)
        // (BEATAMD_GS_NT=32 may fall back to 64-sample tiles per buffer size: one twin only then)
        if (ctx->umax_hist >= 0 && ctx->umax_hist_cg == CG && a.nt == 64) {
            // (a tight fit: whole workgroups per CU are at stake -- 3 instead of 2 at 128 chains and
            // 45 rows -- and a batch that exceeds it only costs that step the full-size twin)
            const int want = ctx->umax_hist + 1;
            if (want < ucap) ucap_fit = std::max(want, 2);
        }
        if (!ctx->umax_pending) {
            BA_HIP(hipMemcpyAsync(ctx->h_umax, ga.umax, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            BA_HIP(hipEventRecord(ctx->umax_event, ctx->stream));
            ctx->umax_pending = true;
            ctx->umax_hist_cg = CG;
        }
    }
    // (fitted twin first, then the full-size twin; a single unguarded launch otherwise)
    const int full_ucap = ucap;
    const int64_t nblocks0 = nblocks;
    const int nt0 = a.nt, ntile0 = a.ntile;
    ScopedTimer tm(ctx, "gfstack");   // (both twins: one timing)
    for (int twin = (ucap_fit ? 1 : 0); twin <= (ucap_fit ? 2 : 0); twin++) {
    ucap = (twin == 1) ? ucap_fit : full_ucap;
    nblocks = nblocks0;
    a.nt = nt0; a.ntile = ntile0;
    a.ucap = ucap;
    a.guard_mode = twin;
    a.guard_fit = ucap_fit;
    a.guard_umax = ga.umax;
    size_t lds = (size_t)ucap * (a.nt + 2) * sizeof(double);
    {
        // two row buffers (the candidates were chosen so that they fit); BEATAMD_GS_DMA=1: the ds_read_b128 layout (A/B)
        BA_CHECK(2 * lds <= 158 * 1024, BEATAMD_EINVAL, "internal: k_gfstack_dma row buffers exceed LDS");
        a.dma = GfKnobs::is(kn.gs_dma, 1) ? 1 : 2;   // 2: ds_read_b64 / pitch NT+1 (default); 1: b128 / pitch NT+2
        if (a.nt == 32 && a.dma != 2) { a.nt = 64; a.ntile = (int)((L.N + 63) / 64); nblocks = ngroups * L.T * a.ntile;
                                        lds = (size_t)ucap * (a.nt + 2) * sizeof(double); }
        BA_CHECK(CG != 1024 || a.dma == 2, BEATAMD_EINVAL, "gfstack: 1024-chain groups need the LDS-DMA kernel");
        BA_CHECK(!ga.windowed || a.dma == 2, BEATAMD_EINVAL, "internal: window slots need the ds_read_b64 kernel");
        a.ws = 0;
        a.nthint = GfKnobs::set(kn.gs_nthint) ? (kn.gs_nthint != 0) : (GS_NTHINT_DEFAULT && ngroups == 1);
        a.f32pair = 0;
        if (a.dma == 2 && a.nt == 64 && CG <= 512 && f32) {
            a.f32pair = 1;
            lds = std::max<size_t>((size_t)ucap * (a.nt + 2) * sizeof(float) * 2, 64 * sizeof(double));
        } else {
            lds *= 2;
        }
    }
    {
        // chain groups of one (target, tile) on one XCD (several groups only)
        a.xcd_order = (ngroups > 1 && !GfKnobs::is(kn.gs_order, 0)) ? 1 : 0;
        if (GfKnobs::is(kn.gs_order, 1)) a.xcd_order = 1;
        if (a.xcd_order) nblocks = ((L.T * a.ntile + 7) / 8) * 8 * ngroups;
    }
    if (a.f32pair)
        snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "k_gfstack_dmaf<%d,%d,%d>", CG / 64, nrow, k.mode);
    else
        snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "%s<%d,%d,%d,%d,%d>",
                 "k_gfstack_dma", CG / 64, nrow, k.mode, a.nt, a.dma == 2 ? 1 : 0);
    ctx->gs_ngtp = GTP;
    ctx->gs_trep = (double)L.T / (double)Ttab;
    ctx->gs_N = L.N;
    ctx->gs_cg = CG;
    ctx->gs_nvar = k.nvar;
    ctx->gs_has_passes = false;
    snprintf(ctx->gf_plan, sizeof(ctx->gf_plan),
             "lane <-> chain kernel with %d-chain groups (%s): row buffers of %d slots%s", CG,
             k.C < 384 ? "small batch" : nrow == 4 ? "multilinear below 192 chains or an odd sample count" : "group size measured fastest",
             ucap, twin ? " sized by the previous launch's distinct-row count" : "");
    {
        dim3 grid((unsigned)nblocks);
        if (CG == 1024) BA_TRY(launch_shared_nrow<16>(nrow, k.mode, grid, lds, ctx->stream, a));
        else if (CG == 512) BA_TRY(launch_shared_nrow<8>(nrow, k.mode, grid, lds, ctx->stream, a));
        else if (CG == 256) BA_TRY(launch_shared_nrow<4>(nrow, k.mode, grid, lds, ctx->stream, a));
        else if (CG == 128) BA_TRY(launch_shared_nrow<2>(nrow, k.mode, grid, lds, ctx->stream, a));
        else BA_TRY(launch_shared_nrow<1>(nrow, k.mode, grid, lds, ctx->stream, a));
    }
    BA_HIP(hipGetLastError());
    }
    BA_HIP(hipGetLastError());
    if (k.mode == GF_RESID_SCALAR) BA_TRY(launch_sum_tiles(ctx, a.partial, k.C * L.T, a.ntile, k.quad));
    return BEATAMD_OK;
}

}  // namespace beatamd
