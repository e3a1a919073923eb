// gfstack.hip -- Green's-function stacking (SeismicGFLibrary.stack_all) for gfx950,
// batched over chains, with the residual / scalar-covariance misfit fused in the epilogue.
//
// Reference arithmetic: beat/ffi/base.py:486-568 (time -> index maps), :607-709
// (stack_all: gather by (duration, starttime) index + slip-weighted sum over patches).
//
// Roofline: HBM.  Per chain-step the kernel must read T*P rows of N doubles (x4 for
// multilinear) exactly once: 0.25 flop/byte.  Design:
//   k_gf_tables  one thread per (chain, target, patch): time -> int16 grid index with
//                numpy semantics (rint = round-half-even, ceil, negative-index wrap),
//                emits a uint32 row id (and the 4 bilinear factors for multilinear).
//   k_gfstack    one 256-thread workgroup per (chain, target, 4 KB-wide sample tile).
//                The row id and the slip are wave-uniform -> fetched with scalar loads
//                (s_load), the row base lives in SGPRs, and every lane streams 16 B
//                (global_load_dwordx4) per row: 1 KB per wave-instruction, 4 KB contiguous
//                per workgroup per row, 8 rows in flight per lane.  fp64 FMA accumulate in
//                registers, no atomics, no LDS traffic on the streaming path (the stream is
//                read exactly once, staging it in LDS would only add latency); LDS is used
//                for the block reduction of the fused misfit.
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace beatamd {

// ------------------------------------------------------------------------ tables
struct TabArgs {
    int interp;
    int64_t C, T, P, D, S;
    double st_min, st_dt, du_min, du_dt;
    ChainVec durations;
    StartTimeSrc st;
    uint32_t *rowoff;  // nn: [C,T,P]   ml: [C,T,P,4] (cc, fc, cf, ff)
    double *fac;       // ml: [C,T,P,4]
    int *status;
    int64_t R;         // patch split: slot t = (real slot t / R, patch range t % R), P = patches per range
};

// numpy: float64 -> int16 astype (wraps modulo 2^16 for in-range int64 values)
__device__ __forceinline__ int to_int16(double x) { return (int)(int16_t)(long long)x; }

// python negative-index wrap; returns false if outside [-n, n)
__device__ __forceinline__ bool wrap_index(int &i, int n)
{
    if (i < 0) i += n;
    return i >= 0 && i < n;
}

__global__ void __launch_bounds__(256) k_gf_tables(TabArgs a)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // a.T = number of table slots: all targets, the distinct station shifts (a.st.shift_off then holds the shift
    // variable of the SLOT), or 1 when the start times do not depend on the target.  Row ids are those of "target" =
    // slot index; the stacking kernels add (target - slot) x rows per target
    const int64_t total = a.C * a.T * a.P;
    if (idx >= total) return;
    const int64_t p = idx % a.P;
    const int64_t ct = idx / a.P;
    const int64_t t = ct % a.T;
    const int64_t c = ct / a.T;

    // (patch split: the slot is (real slot ts, range r), the patch is r * P + p of the real model)
    const int64_t ts = t / a.R, pr = (t % a.R) * a.P + p, Preal = a.P * a.R;
    double st;
    if (a.st.explicit_st) {
        st = a.st.explicit_st[(c * (a.T / a.R) + ts) * Preal + pr];
    } else {
        // seismic.py:1283-1296: tile(starttimes0, T) - repeat(time_shifts[station_idx], P)
        st = a.st.starttimes0[c * Preal + pr];
        if (a.st.shift_off) st = st - a.st.Q[c * a.st.nparams + a.st.shift_off[ts]];
    }
    const double du = a.durations.base[c * a.durations.stride + a.durations.off + pr];
    const int D = (int)a.D, S = (int)a.S;
    const int64_t row0 = (t * a.P + p) * a.D;  // row id = (row0 + di) * S + si
    bool ok = true;
    if (a.interp == BEATAMD_NEAREST_NEIGHBOR) {
        // base.py:506-511 / :553-558: round((x - min) / sampling).astype(int16)
        int si = to_int16(rint((st - a.st_min) / a.st_dt));
        int di = to_int16(rint((du - a.du_min) / a.du_dt));
        const bool ok_s = wrap_index(si, S), ok_d = wrap_index(di, D);
        ok = ok_s && ok_d;
        if (!ok) { si = 0; di = 0; }
        a.rowoff[idx] = (uint32_t)((row0 + di) * a.S + si);
    } else {
        // base.py:512-517 / :559-564: ceil -> int16, factor = ceil - x (weight of the FLOOR node)
        const double ds = (st - a.st_min) / a.st_dt;
        const double dd = (du - a.du_min) / a.du_dt;
        int sc = to_int16(ceil(ds)), dc = to_int16(ceil(dd));
        const double stf = (double)sc - ds, rtf = (double)dc - dd;
        int sf = sc - 1, df = dc - 1;
        const bool o1 = wrap_index(sc, S), o2 = wrap_index(dc, D), o3 = wrap_index(sf, S),
                   o4 = wrap_index(df, D);
        ok = o1 && o2 && o3 && o4;
        if (!ok) { sc = dc = sf = df = 0; }
        uint32_t *ro = a.rowoff + idx * 4;
        double *fa = a.fac + idx * 4;
        ro[0] = (uint32_t)((row0 + dc) * a.S + sc);  // st ceil , rt ceil
        ro[1] = (uint32_t)((row0 + dc) * a.S + sf);  // st floor, rt ceil
        ro[2] = (uint32_t)((row0 + df) * a.S + sc);  // st ceil , rt floor
        ro[3] = (uint32_t)((row0 + df) * a.S + sf);  // st floor, rt floor
        // base.py:676-679 (the slip factor is applied in k_gfstack)
        fa[0] = (1 - stf) * (1 - rtf);
        fa[1] = stf * (1.0 - rtf);
        fa[2] = (1 - stf) * rtf;
        fa[3] = stf * rtf;
    }
    if (!ok) {
        atomicOr(a.status, ST_INDEX_OOB);
        if (a.st.chain_bad) a.st.chain_bad[c] = 1;
    }
}

// ------------------------------------------------------------------------ stacking
struct GfArgs {
    const double *G[4];
    ChainVec slips[4];
    int64_t T, P, N;
    const uint32_t *rowoff;
    const double *fac;
    const double *data;
    const double *wscalar;
    double *out;
    double *partial;  // [C*T, ntile]
    int ntile;
    int order;        // 0: blocks ordered (chain, target, tile); 1: (group, target, chain, tile)
    int64_t C;
    int cgroup;       // order 1: chains per group
    int64_t Ttab, rows_per_target;   // tables per (chain, table slot, patch): Ttab slots (T, the station shifts, or 1)
    const int32_t *tslot;            // [T] slot of a target (nullptr: Ttab == 1 ? 0 : t)
    // stand-in launch behind k_gfstack_runs: works only when *guard != 0 (the runs kernel's tables overflowed); a small
    // grid whose workgroups then loop over the nblocks tiles (round 6: a full grid of workgroups that only read the flag
    // cost 35-100 us per launch on configs[3] with 120-sample traces)
    const int *guard;
    int64_t nblocks;
    int64_t R;        // patch split: target t = (real target, range t % R): slips at patch (t % R) * P + p
};

template <int W> struct VecT;
template <> struct VecT<2> { using type = double2; };
template <> struct VecT<1> { using type = double; };

template <int W>
__device__ __forceinline__ typename VecT<W>::type ldg(const double *p)
{
    return *reinterpret_cast<const typename VecT<W>::type *>(p);
}
__device__ __forceinline__ void fma_acc(double2 &acc, double2 x, double w)
{
    acc.x = fma(x.x, w, acc.x);
    acc.y = fma(x.y, w, acc.y);
}
__device__ __forceinline__ void fma_acc(double &acc, double x, double w) { acc = fma(x, w, acc); }

// INTERP 0 nn / 1 ml ; NVAR slip variables ; VEC chunks of 256*W samples per thread ;
// W doubles per lane per load (2 when N is even -> 16-byte loads) ; MODE GfMode
template <int INTERP, int NVAR, int VEC, int W, int MODE>
__device__ __forceinline__ void gfstack_tile(const GfArgs &a, const int64_t bid)
{
    using V = typename VecT<W>::type;
    constexpr int NROW = INTERP ? 4 : 1;
    constexpr int CH = 256 * W;  // samples per chunk
    // rows of V in flight per lane ~ 8
    constexpr int U = (8 / (NROW * NVAR * VEC)) > 0 ? (8 / (NROW * NVAR * VEC)) : 1;

    const int tile = (int)(bid % a.ntile);
    const int64_t bq = bid / a.ntile;
    int64_t c, t;
    if (a.order == 0) {
        c = bq / a.T;
        t = bq - c * a.T;
    } else {
        // target-major: the chains of one target run concurrently and walk the patches
        // roughly in step, so rows shared by several chains are served from L2 / MALL
        // within a group of `cgroup` chains
        const int64_t per_group = (int64_t)a.cgroup * a.T;
        const int64_t g = bq / per_group;
        const int64_t r = bq - g * per_group;
        const int64_t c0 = g * a.cgroup;
        const int64_t gsz = min((int64_t)a.cgroup, a.C - c0);  // last group may be short
        // groups before the last are full, so r indexes (t, c) with the group's own size
        t = r / gsz;
        c = c0 + (r - t * gsz);
    }
    const int64_t ct = c * a.T + t;
    const int64_t slot = a.tslot ? (int64_t)a.tslot[t] : (a.Ttab == 1 ? 0 : t);
    const int64_t ctt = c * a.Ttab + slot;                                    // table cell
    const int64_t tbase = (t - slot) * a.rows_per_target;                     // rows
    const int64_t N = a.N;
    const int P = (int)a.P;

    int64_t n[VEC];
    bool inb[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        const int64_t nn = ((int64_t)tile * VEC + v) * CH + (int64_t)threadIdx.x * W;
        inb[v] = nn < N;  // N even when W == 2 -> a pair is never split
        n[v] = inb[v] ? nn : 0;  // out-of-range lanes re-read sample 0 (discarded)
    }
    const uint32_t *ro = a.rowoff + ctt * a.P * NROW;
    const double *fa = INTERP ? (a.fac + ctt * a.P * NROW) : nullptr;
    const double *sl[NVAR];
#pragma unroll
    for (int iv = 0; iv < NVAR; iv++)
        sl[iv] = a.slips[iv].base + c * a.slips[iv].stride + a.slips[iv].off + (t % a.R) * a.P;

    V acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = V{};

    int p = 0;
    for (; p + U <= P; p += U) {
        V x[U][NROW][NVAR][VEC];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < NROW; k++) {
                const int64_t roff = ((int64_t)ro[(p + u) * NROW + k] + tbase) * N;  // wave-uniform (SGPR)
#pragma unroll
                for (int iv = 0; iv < NVAR; iv++) {
                    const double *row = a.G[iv] + roff;
#pragma unroll
                    for (int v = 0; v < VEC; v++) x[u][k][iv][v] = ldg<W>(row + n[v]);
                }
            }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < NROW; k++)
#pragma unroll
                for (int iv = 0; iv < NVAR; iv++) {
                    double w = sl[iv][p + u];
                    if (INTERP) w = fa[(p + u) * NROW + k] * w;
#pragma unroll
                    for (int v = 0; v < VEC; v++) fma_acc(acc[v], x[u][k][iv][v], w);
                }
    }
    for (; p < P; p++) {
#pragma unroll
        for (int k = 0; k < NROW; k++) {
            const int64_t roff = ((int64_t)ro[p * NROW + k] + tbase) * N;
#pragma unroll
            for (int iv = 0; iv < NVAR; iv++) {
                double w = sl[iv][p];
                if (INTERP) w = fa[p * NROW + k] * w;
                const double *row = a.G[iv] + roff;
#pragma unroll
                for (int v = 0; v < VEC; v++) fma_acc(acc[v], ldg<W>(row + n[v]), w);
            }
        }
    }

    // ---- epilogue
    if (MODE == GF_STORE_SYN) {
#pragma unroll
        for (int v = 0; v < VEC; v++)
            if (inb[v]) *reinterpret_cast<V *>(a.out + ct * N + n[v]) = acc[v];
    } else if (MODE == GF_RESID_STORE) {
        // seismic.py:1332: residuals = data - synthetics
#pragma unroll
        for (int v = 0; v < VEC; v++)
            if (inb[v]) {
                V d = ldg<W>(a.data + t * N + n[v]);
                V r;
                if constexpr (W == 2) { r.x = d.x - acc[v].x; r.y = d.y - acc[v].y; }
                else { r = d - acc[v]; }
                *reinterpret_cast<V *>(a.out + ct * N + n[v]) = r;
            }
    } else {
        // distributions.py:128-136 with W = w I: tmp = w * r ; tmp . tmp
        const double w = a.wscalar[t];
        double q = 0.0;
#pragma unroll
        for (int v = 0; v < VEC; v++)
            if (inb[v]) {
                V d = ldg<W>(a.data + t * N + n[v]);
                if constexpr (W == 2) {
                    double t0 = w * (d.x - acc[v].x), t1 = w * (d.y - acc[v].y);
                    q = fma(t0, t0, q);
                    q = fma(t1, t1, q);
                } else {
                    double t0 = w * (d - acc[v]);
                    q = fma(t0, t0, q);
                }
            }
        for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
        __shared__ double red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0)
            a.partial[ct * a.ntile + tile] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

template <int INTERP, int NVAR, int VEC, int W, int MODE>
__global__ void __launch_bounds__(256) k_gfstack(GfArgs a)
{
    if (a.guard) {
        if (*a.guard == 0) return;
        for (int64_t b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
            gfstack_tile<INTERP, NVAR, VEC, W, MODE>(a, b);
            __syncthreads();
        }
    } else {
        gfstack_tile<INTERP, NVAR, VEC, W, MODE>(a, blockIdx.x);
    }
}

// float-storage copy of a library: g32 = (float)g and g itself rounded to the same values, so that
// every kernel -- whichever copy it reads -- sees one library
__global__ void __launch_bounds__(256) k_round_to_f32(double *g, float *g32, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float f = (float)g[i];
        g32[i] = f;
        g[i] = (double)f;
    }
}

int launch_round_to_f32(beatamd_ctx *ctx, double *g, float *g32, int64_t n)
{
    if (n == 0) return BEATAMD_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(k_round_to_f32, dim3(grid), dim3(256), 0, ctx->stream, g, g32, n);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// guard (nullable): the launch works only when (*guard != 0) == (want != 0) -- the two producers of a misfit, the runs
// kernel and its stand-in, each bring their own tile sums
__global__ void __launch_bounds__(256) k_sum_tiles(const double *partial, int64_t n, int ntile,
                                                  double *quad, const int *guard, int want)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (guard && (*guard != 0) != (want != 0)) return;
    double s = 0.0;
    for (int k = 0; k < ntile; k++) s += partial[i * ntile + k];  // fixed order: deterministic
    quad[i] = s;
}

int launch_sum_tiles(beatamd_ctx *ctx, const double *partial, int64_t n, int ntile, double *quad, const int *guard, int want)
{
    hipLaunchKernelGGL(k_sum_tiles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       partial, n, ntile, quad, guard, want);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

// mode 3 (bidiagonal whitening operator): the tiles of a (chain, target) in ascending order, behind each tile's inner
// samples its last one -- y = W[i,i] r_last(tile) + W[i,i+1] r_first(tile + 1), the two products of k_quadform_banded<1> --
// fixed order: deterministic, the same on every rank
__global__ void __launch_bounds__(256) k_sum_tiles_band1(const double *partial, const double *edges, const double *band_w,
                                                        int64_t n, int64_t T, int64_t N, int ntile, int NT, double *quad,
                                                        const int *guard, int want)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (guard && (*guard != 0) != (want != 0)) return;
    const int64_t t = i % T;
    double s = 0.0;
    for (int k = 0; k < ntile; k++) {
        s += partial[i * ntile + k];
        if (k + 1 < ntile) {
            const int64_t smp = (int64_t)k * NT + NT - 1;
            const double *w = band_w + (t * N + smp) * 2;
            double y = fma(w[0], edges[(i * ntile + k) * 2 + 1], 0.0);
            y = fma(w[1], edges[(i * ntile + k + 1) * 2], y);
            s = fma(y, y, s);
        }
    }
    quad[i] = s;
}

int launch_sum_tiles_band1(beatamd_ctx *ctx, const double *partial, const double *edges, const double *band_w, int64_t C,
                           int64_t T, int64_t N, int ntile, int NT, double *quad, const int *guard, int want)
{
    const int64_t n = C * T;
    hipLaunchKernelGGL(k_sum_tiles_band1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, partial, edges, band_w,
                       n, T, N, ntile, NT, quad, guard, want);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

template <int INTERP, int NVAR, int VEC, int W>
static void launch_mode(int mode, dim3 grid, hipStream_t s, const GfArgs &a)
{
    if (mode == GF_STORE_SYN)
        hipLaunchKernelGGL((k_gfstack<INTERP, NVAR, VEC, W, GF_STORE_SYN>), grid, dim3(256), 0, s, a);
    else if (mode == GF_RESID_SCALAR)
        hipLaunchKernelGGL((k_gfstack<INTERP, NVAR, VEC, W, GF_RESID_SCALAR>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_gfstack<INTERP, NVAR, VEC, W, GF_RESID_STORE>), grid, dim3(256), 0, s, a);
}

template <int INTERP, int VEC, int W>
static void launch_nvar(int nvar, int mode, dim3 grid, hipStream_t s, const GfArgs &a)
{
    if (nvar == 1) launch_mode<INTERP, 1, VEC, W>(mode, grid, s, a);
    else if (nvar == 2) launch_mode<INTERP, 2, VEC, W>(mode, grid, s, a);
    else launch_mode<INTERP, 3, VEC, W>(mode, grid, s, a);
}

static int launch_gfstack_impl(beatamd_ctx *ctx, const GfStackCall &call);

// ---- patch split of small-N libraries --------------------------------------------------------------------------------
// A (target, 64-sample tile) walk over P patches is a SERIAL path of P (x slip variables) steps per workgroup; a library
// of short traces -- the reference's realistic case: 60 s at 2 Hz = 120 samples, SURVEY 8(d) config 4 -- has T * ceil(N / 64)
// = 70 of them for 256 CUs, each fetching its index tables cold.  Such libraries are stacked in R patch RANGES: the
// library [T, P, D, S, N] viewed as [T*R, P/R, D, S, N] (the same memory), R times as many and R times shorter walks, the
// ranges' partial synthetics summed in range order by k_split_combine, which also carries the epilogue.  R depends on the
// library's shape ONLY (never on the batch): a chain's result cannot depend on the batch it is in; every stacking kernel
// takes the view unchanged (tables per virtual slot), so the kernels stay bitwise equal to each other.
// Rule: N <= 256 and at least 32 patches per range; among the divisors R of P (<= 32) the one that minimises
//   ceil(walks * R / CUs) * (P / R + 5)
// -- the stacking kernels keep one workgroup per CU, a walk costs its steps plus ~5 steps of prologue / epilogue (measured on
// configs[3] with 120 samples, 70 walks x 400 patches: R = 4: 1.00, 5: 0.80, 8: 0.82, 10: 0.69, 16: 0.76, 25: 0.72 ms).
int gf_patch_ranges(int64_t T, int64_t P, int64_t N, int num_cu)
{
    if (N > 256 || P < 64) return 1;
    const int64_t walks = T * ((N + 63) / 64);
    const int64_t ncu = std::max(1, num_cu);
    if (walks >= 2 * ncu) return 1;
    int best = 1;
    int64_t best_cost = ((walks + ncu - 1) / ncu) * (P + 5);
    for (int d = 2; d <= 32 && P / d >= 32; d++) {
        if (P % d) continue;
        const int64_t cost = ((walks * d + ncu - 1) / ncu) * (P / d + 5);
        if (cost < best_cost) { best = d; best_cost = cost; }
    }
    return best;
}

static int gf_patch_split(const SeisLib &L, const GfKnobs &kn, int num_cu)
{
    const int knob = GfKnobs::get(kn.gf_split, -1);
    if (knob == 0 || knob == 1) return 1;
    if (knob > 1) return (L.P % knob == 0) ? knob : 1;
    return gf_patch_ranges(L.T, L.P, L.N, num_cu);
}

__global__ void __launch_bounds__(256) k_split_tslot(int64_t Tv, int R, const int32_t *tslot, int32_t *out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < Tv) out[t] = (tslot ? tslot[t / R] : 0) * R + (int32_t)(t % R);
}

// syn[c,t,n] = sum over the ranges r ascending of part[c, t*R + r, n], then the epilogue of `mode`:
//   0 out = syn   2 / 3 out = d - syn (seismic.py:1332)   1 quad[c,t] = sum_n (w_t (d - syn))^2 -- per 64-sample tile
//   ascending, then the tiles ascending (the order of the lane <-> chain kernels)
template <int MODE>
__global__ void __launch_bounds__(256) k_split_combine(const double *part, int64_t C, int64_t T, int64_t N, int R,
                                                      const double *data, const double *wscalar, double *out, double *quad,
                                                      const double *band_w)
{
    __shared__ double sq[256 + 1];
    __shared__ double tsum[4];
    const int64_t ct = blockIdx.x, t = ct % T, c = ct / T;
    const int n = threadIdx.x;
    double v = 0.0;
    if (n < N) {
        const double *pp = part + ((c * T + t) * R) * N + n;
        double syn = pp[0];
        for (int r = 1; r < R; r++) syn += pp[(int64_t)r * N];
        if (MODE == GF_STORE_SYN) out[ct * N + n] = syn;
        else if (MODE == GF_RESID_SCALAR) v = wscalar[t] * (data[t * N + n] - syn);
        else if (MODE == GF_RESID_BAND1) v = data[t * N + n] - syn;              // seismic.py:1332
        else out[ct * N + n] = data[t * N + n] - syn;
    }
    if (MODE != GF_RESID_SCALAR && MODE != GF_RESID_BAND1) return;
    sq[n] = v;
    __syncthreads();
    const int ntile = (int)((N + 63) / 64);
    if (MODE == GF_RESID_BAND1) {
        // the bidiagonal misfit in the canonical order of quadform.hip (k_quadform_band1): per 64-sample tile the samples
        // but its last, ascending, + the trace's very last sample; then the tiles ascending with every tile's boundary term
        // (round 6: no residual store, no second kernel behind the patch ranges either)
        // (the band rows of the target through LDS: read from memory inside the serial loops they were a round trip per
        // sample)
        __shared__ double w[2 * 256];
        if (n < N) {
            w[2 * n] = band_w[(t * N + n) * 2];
            w[2 * n + 1] = band_w[(t * N + n) * 2 + 1];
        }
        __syncthreads();
        if (n < ntile) {
            const int n0 = n * 64, nv = (int)min((int64_t)64, N - (int64_t)n0);
            double q = 0.0, ri = sq[n0];
            for (int i = 0; i + 1 < nv; i++) {
                const double rn = sq[n0 + i + 1];
                double y = fma(w[2 * (n0 + i)], ri, 0.0);
                y = fma(w[2 * (n0 + i) + 1], rn, y);
                q = fma(y, y, q);
                ri = rn;
            }
            if (n0 + nv == N) {
                const double y = fma(w[2 * (N - 1)], ri, 0.0);
                q = fma(y, y, q);
            }
            tsum[n] = q;
        }
        __syncthreads();
        if (n == 0) {
            double s_ = 0.0;
            for (int k = 0; k < ntile; k++) {
                s_ += tsum[k];
                if (k + 1 < ntile) {
                    const int last = k * 64 + 63;
                    double y = fma(w[2 * last], sq[last], 0.0);
                    y = fma(w[2 * last + 1], sq[last + 1], y);
                    s_ = fma(y, y, s_);
                }
            }
            quad[ct] = s_;
        }
        return;
    }
    if (n < ntile) {
        double q = 0.0;
        const int hi = (int)min((int64_t)64, N - (int64_t)n * 64);
        for (int i = 0; i < hi; i++) q = fma(sq[n * 64 + i], sq[n * 64 + i], q);
        tsum[n] = q;
    }
    __syncthreads();
    if (n == 0) {
        double s_ = 0.0;
        for (int k = 0; k < ntile; k++) s_ += tsum[k];
        quad[ct] = s_;
    }
}

static int launch_gfstack_split(beatamd_ctx *ctx, const GfStackCall &call, int R)
{
    const SeisLib &L = *call.libs[0];
    SeisLib views[4];
    GfStackCall v = call;
    for (int i = 0; i < call.nvar; i++) {
        views[i] = *call.libs[i];
        views[i].T = L.T * R;
        views[i].P = L.P / R;
        v.libs[i] = &views[i];
    }
    v.patch_split = R;
    v.mode = GF_STORE_SYN;
    void *p = nullptr;
    BA_TRY(ctx->get_scratch(SL_SPLIT, (size_t)call.C * L.T * R * L.N * sizeof(double), &p));
    double *part = (double *)p;
    v.out = part;
    v.quad = nullptr; v.data = nullptr; v.wscalar = nullptr; v.band_w = nullptr;
    BA_TRY(launch_gfstack_impl(ctx, v));
    for (int i = 0; i < call.nvar; i++) { views[i].g = nullptr; views[i].g32 = nullptr; }   // (views own nothing)
    {
        // (kernel name / plan of the stacking launch stay; the plan says that the library was split)
        const size_t n0 = strlen(ctx->gf_plan);
        snprintf(ctx->gf_plan + n0, sizeof(ctx->gf_plan) - n0, "; %lld-sample traces: patches stacked in %d ranges of %lld "
                 "(%lld walks instead of %lld), partial synthetics summed in range order", (long long)L.N, R, (long long)(L.P / R),
                 (long long)(L.T * R * ((L.N + 63) / 64)), (long long)(L.T * ((L.N + 63) / 64)));
    }
    const dim3 grid((unsigned)(call.C * L.T));
    BA_CHECK(call.C * L.T < (int64_t)0x7fffffff, BEATAMD_EINVAL, "gfstack: batch too large");
    ScopedTimer tm(ctx, "gfcombine");
    switch (call.mode) {
    case GF_STORE_SYN:
        hipLaunchKernelGGL(k_split_combine<GF_STORE_SYN>, grid, dim3(256), 0, ctx->stream, part, call.C, L.T, L.N, R, call.data,
                           call.wscalar, call.out, call.quad, (const double *)nullptr);
        break;
    case GF_RESID_SCALAR:
        hipLaunchKernelGGL(k_split_combine<GF_RESID_SCALAR>, grid, dim3(256), 0, ctx->stream, part, call.C, L.T, L.N, R, call.data,
                           call.wscalar, call.out, call.quad, (const double *)nullptr);
        break;
    case GF_RESID_BAND1:
        if (GfKnobs::get(gf_knobs(ctx).qf_fuse, 1) != 0) {
            hipLaunchKernelGGL(k_split_combine<GF_RESID_BAND1>, grid, dim3(256), 0, ctx->stream, part, call.C, L.T, L.N, R, call.data,
                               call.wscalar, call.out, call.quad, call.band_w);
            ctx->gf_band_fused = true;
            break;
        }
        [[fallthrough]];
    default:
        hipLaunchKernelGGL(k_split_combine<GF_RESID_STORE>, grid, dim3(256), 0, ctx->stream, part, call.C, L.T, L.N, R, call.data,
                           call.wscalar, call.out, call.quad, (const double *)nullptr);
    }
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

int launch_gfstack(beatamd_ctx *ctx, const GfStackCall &call)
{
    const int R = call.libs[0] ? gf_patch_split(*call.libs[0], gf_knobs(ctx), ctx->num_cu) : 1;
    if (R > 1) {
        ctx->gf_band_fused = false;
        BA_TRY(launch_gfstack_split(ctx, call, R));
        if (call.mode == GF_RESID_BAND1 && !ctx->gf_band_fused) {
            const SeisLib &L = *call.libs[0];
            BA_TRY(launch_quadform_banded(ctx, call.band_w, 1, L.N, L.T, call.C, call.out, L.T * L.N, L.N, call.quad, L.T));
        }
        return BEATAMD_OK;
    }
    if (call.mode != GF_RESID_BAND1) return launch_gfstack_impl(ctx, call);
    // bidiagonal whitening operator: fused into the stacking kernel where that kernel has the epilogue (k_gfstack_ws),
    // else residual store + k_quadform_banded -- the caller gets quad [C,T] either way
    BA_CHECK(call.band_w && call.quad && call.out && call.data, BEATAMD_EINVAL, "gfstack: mode 3 needs band_w, quad, out, data");
    ctx->gf_band_fused = false;
    BA_TRY(launch_gfstack_impl(ctx, call));
    if (!ctx->gf_band_fused) {
        const SeisLib &L = *call.libs[0];
        BA_TRY(launch_quadform_banded(ctx, call.band_w, 1, L.N, L.T, call.C, call.out, L.T * L.N, L.N, call.quad, L.T));
    }
    return BEATAMD_OK;
}

static int launch_gfstack_impl(beatamd_ctx *ctx, const GfStackCall &call)
{
    GfStackCall k = call;
    const GfKnobs &kn = gf_knobs(ctx);
    k.knobs = &kn;
    const SeisLib &L = *k.libs[0];
    BA_CHECK(k.nvar >= 1 && k.nvar <= 3, BEATAMD_EINVAL, "gfstack: 1..3 slip variables supported");
    for (int v = 1; v < k.nvar; v++) {
        const SeisLib &M = *k.libs[v];
        BA_CHECK(M.T == L.T && M.P == L.P && M.D == L.D && M.S == L.S && M.N == L.N &&
                     M.st_min == L.st_min && M.st_dt == L.st_dt && M.du_min == L.du_min &&
                     M.du_dt == L.du_dt,
                 BEATAMD_EINVAL, "gfstack: libraries of the slip variables differ in shape/grid");
    }
    BA_CHECK(L.T * L.P * L.D * L.S < (int64_t)0xffffffffLL, BEATAMD_EINVAL,
             "gfstack: library has more than 2^32 rows");
    if (k.C == 0) return BEATAMD_OK;
    // Without explicit start times and without station shifts the start-time and duration
    // indices of a chain are the same for every target (seismic.py:1283-1296 tiles starttimes0
    // over the targets): the tables are then built once per (chain, patch) instead of T times.
    const bool tinv = !k.st.explicit_st && !k.st.shift_off && !GfKnobs::is(kn.gf_tinv, 0);
    // ... and with station corrections only on the station: one table slot per distinct shift variable (the channels of
    // a station share it)
    const bool slots = !k.st.explicit_st && k.st.shift_off && k.st.nslot > 0 && !GfKnobs::is(kn.gf_tinv, 0);
    // patch split (L is the VIEW [T*R, P/R, ...]): every table slot once per patch range -- R slots without station
    // shifts, nslot * R with them; virtual target t*R + r uses slot (slot of t)*R + r
    const int64_t R = k.patch_split;
    const int64_t Ttab = tinv ? R : slots ? (int64_t)k.st.nslot * R : L.T;
    k.tslot = slots ? k.st.tslot : nullptr;
    void *p = nullptr;
    if (R > 1 && (tinv || slots)) {
        BA_TRY(ctx->get_scratch(SL_TSLOT, (size_t)L.T * sizeof(int32_t), &p));
        hipLaunchKernelGGL(k_split_tslot, dim3((unsigned)((L.T + 255) / 256)), dim3(256), 0, ctx->stream, L.T, (int)R,
                           slots ? k.st.tslot : nullptr, (int32_t *)p);
        k.tslot = (const int32_t *)p;
    }
    const int64_t CTP = k.C * Ttab * L.P;
    const int nrow = k.interp == BEATAMD_MULTILINEAR ? 4 : 1;

    TabArgs ta;
    ta.interp = k.interp;
    ta.R = R;
    ta.C = k.C; ta.T = Ttab; ta.P = L.P; ta.D = L.D; ta.S = L.S;
    ta.st_min = L.st_min; ta.st_dt = L.st_dt; ta.du_min = L.du_min; ta.du_dt = L.du_dt;
    ta.durations = k.durations;
    ta.st = k.st;
    if (slots) ta.st.shift_off = k.st.slot_shift_off;
    ta.status = ctx->d_status;
    BA_TRY(ctx->get_scratch(SL_ROWOFF, (size_t)CTP * nrow * sizeof(uint32_t), &p));
    ta.rowoff = (uint32_t *)p;
    ta.fac = nullptr;
    if (nrow == 4) {
        BA_TRY(ctx->get_scratch(SL_WEIGHTS, (size_t)CTP * 4 * sizeof(double), &p));
        ta.fac = (double *)p;
    }
    {
        ScopedTimer tm(ctx, "tables");
        hipLaunchKernelGGL(k_gf_tables, dim3((unsigned)((CTP + 255) / 256)), dim3(256), 0,
                           ctx->stream, ta);
    }
    BA_HIP(hipGetLastError());

    // float storage requested and every library has its float copy: the lane <-> chain kernels read it (the
    // cell kernel works on the float64 rows)
    bool f32_all = k.f32;
    for (int v = 0; v < k.nvar; v++) f32_all = f32_all && k.libs[v]->g32 != nullptr;
    // multilinear from 192 chains on: the runs kernel (gfcell.hip).  When its tables can overflow (more row passes than they
    // are sized for) the streaming kernel below is enqueued behind it as a stand-in that works only if they did.
    const int *standin = nullptr;
    // (mode 3: launch_gfstack_shared and the runs kernel have the epilogue -- round 6 --, the streaming kernel stores the
    // residuals; as the runs kernel's stand-in it is followed by a guarded k_quadform_band1)
    const int mode_in = k.mode;
    // (BEATAMD_QF_FUSE=0: residual store + k_quadform_band1 behind every kernel -- A/B, tests)
    const bool fuse_runs = mode_in == GF_RESID_BAND1 && GfKnobs::get(kn.qf_fuse, 1) != 0;
    if (mode_in == GF_RESID_BAND1 && !fuse_runs) k.mode = GF_RESID_STORE;
    if (!f32_all && gfstack_ml_applicable(k)) {
        BA_TRY(launch_gfstack_ml(ctx, k, ta.rowoff, ta.fac, Ttab, &standin));
        if (fuse_runs) ctx->gf_band_fused = true;
        if (!standin) return BEATAMD_OK;
    }
    if (mode_in == GF_RESID_BAND1) k.mode = GF_RESID_STORE;
    if (!standin) {
        int cg = 0, ucap = 0;
        if (gfstack_shared_applicable(k, &cg, &ucap)) {
            k.mode = mode_in;
            // Chains per workgroup: which size is fastest depends on the library (distinct rows a
            // group can share, D*S), the batch and the population, so it is MEASURED once per
            // problem shape -- every candidate is launched on the real inputs (the kernels are
            // bitwise equal for every group size, the outputs are simply rewritten) and the
            // fastest is kept.  BEATAMD_GS_CG fixes the size, BEATAMD_GS_TUNE=0 uses the static
            // table of pick_group (measured on config 3).
            const bool tune = !GfKnobs::set(kn.gs_cg) && !GfKnobs::is(kn.gs_tune, 0);
            if (tune) {
                // (key: the batch in whole 512-chain groups, capped -- 4096 and 4100 chains choose alike --; the epilogue
                // mode costs every group size the same and is not part of it)
                const int64_t cbucket = k.C < 512 ? k.C : 512 * std::min<int64_t>((k.C + 511) / 512, 16);
                const std::vector<int64_t> key = {cbucket, nrow, k.nvar, L.T, L.P, L.D, L.S, L.N, Ttab, f32_all ? 1 : 0};
                auto it = ctx->gs_tuned.find(key);
                if (it == ctx->gs_tuned.end()) {
                    int cgs[4], ucaps[4];
                    const int nc = gfstack_shared_candidates(k, cgs, ucaps);
                    int best = -1;
                    float best_ms = 0.f, ms_of[4] = {0.f, 0.f, 0.f, 0.f};
                    hipEvent_t e0, e1;
                    BA_HIP(hipEventCreate(&e0));
                    BA_HIP(hipEventCreate(&e1));
                    for (int i = 0; i < nc && nc > 1; i++) {
                        float ms_min = 0.f;
                        for (int rep = 0; rep < 2; rep++) {   // first launch of a size: warm-up (code, tables' scratch)
                            BA_HIP(hipEventRecord(e0, ctx->stream));
                            BA_TRY(launch_gfstack_shared(ctx, k, ta.rowoff, ta.fac, cgs[i], ucaps[i], Ttab));
                            BA_HIP(hipEventRecord(e1, ctx->stream));
                            BA_HIP(hipEventSynchronize(e1));
                            float ms = 0.f;
                            BA_HIP(hipEventElapsedTime(&ms, e0, e1));
                            if (rep == 1) ms_min = ms;
                        }
                        ms_of[i] = ms_min;
                        if (best < 0 || ms_min < best_ms) { best = i; best_ms = ms_min; }
                        // a candidate twice as slow as the best so far: the smaller sizes behind it only stage more rows
                        if (ms_min > 2.f * best_ms) break;
                    }
                    (void)hipEventDestroy(e0);
                    (void)hipEventDestroy(e1);
                    if (nc == 1) best = 0;
                    if (best >= 0) it = ctx->gs_tuned.emplace(key, std::make_pair(cgs[best], ucaps[best])).first;
                    // what was measured, once per problem shape: beatamd_ctx_gf_tune_log (and stderr under BEATAMD_VERBOSE)
                    {
                        char *w = ctx->gf_tune_log;
                        size_t left = sizeof(ctx->gf_tune_log);
                        int n = snprintf(w, left, "group size for %lld chains (T %lld, P %lld, D*S %lld, N %lld, %d row(s) per chain): ",
                                         (long long)k.C, (long long)L.T, (long long)L.P, (long long)(L.D * L.S), (long long)L.N, nrow);
                        for (int i = 0; i < nc && n > 0 && (size_t)n < left; i++) {
                            w += n; left -= (size_t)n;
                            n = ms_of[i] > 0.f ? snprintf(w, left, "%d: %.3f ms%s", cgs[i], ms_of[i], i + 1 < nc ? ", " : "")
                                               : snprintf(w, left, "%d: %s%s", cgs[i], nc == 1 ? "only candidate" : "not timed", i + 1 < nc ? ", " : "");
                        }
                        if (best >= 0 && n > 0 && (size_t)n < left) { w += n; left -= (size_t)n; snprintf(w, left, " -> %d", cgs[best]); }
                        if (getenv("BEATAMD_VERBOSE")) fprintf(stderr, "beat_amd: %s\n", ctx->gf_tune_log);
                    }
                }
                if (it != ctx->gs_tuned.end()) { cg = it->second.first; ucap = it->second.second; }
            }
            return launch_gfstack_shared(ctx, k, ta.rowoff, ta.fac, cg, ucap, Ttab);
        }
    }

    GfArgs a;
    memset(&a, 0, sizeof(a));
    for (int v = 0; v < k.nvar; v++) {
        a.G[v] = k.libs[v]->g;
        a.slips[v] = k.slips[v];
    }
    a.T = L.T; a.P = L.P; a.N = L.N;
    a.Ttab = Ttab; a.rows_per_target = L.P * L.D * L.S;
    a.tslot = k.tslot;
    a.C = k.C;
    a.R = R;
    {
        a.order = GfKnobs::get(kn.gf_order, 1);
        a.cgroup = GfKnobs::get(kn.gf_cgroup, 128);
        if (a.cgroup < 1) a.cgroup = 1;
    }
    a.rowoff = ta.rowoff;
    a.fac = ta.fac;
    a.data = k.data;
    a.wscalar = k.wscalar;
    a.out = k.out;
    a.guard = standin;
    const int W = (L.N % 2 == 0) ? 2 : 1;
    const int VEC = 1;
    const int64_t tile_w = (int64_t)256 * W * VEC;
    a.ntile = (int)((L.N + tile_w - 1) / tile_w);
    if (k.mode == GF_RESID_SCALAR) {
        // (a stand-in keeps its tile sums apart from those of the kernel it stands in for)
        BA_TRY(ctx->get_scratch(standin ? SL_PARTIAL2 : SL_PARTIAL, (size_t)k.C * L.T * a.ntile * sizeof(double), &p));
        a.partial = (double *)p;
    }
    const int64_t nblocks = k.C * L.T * a.ntile;
    BA_CHECK(nblocks < (int64_t)0x7fffffff, BEATAMD_EINVAL, "gfstack: batch too large (%lld blocks)",
             (long long)nblocks);
    a.nblocks = nblocks;
    dim3 grid((unsigned)(standin ? std::min<int64_t>(nblocks, (int64_t)ctx->num_cu * 8) : nblocks));
    if (!standin) {
        snprintf(ctx->last_gf_kernel, sizeof(ctx->last_gf_kernel), "k_gfstack<%d,%d,%d,%d,%d>",
                 k.interp == BEATAMD_MULTILINEAR ? 1 : 0, k.nvar, VEC, W, k.mode);
        snprintf(ctx->gf_plan, sizeof(ctx->gf_plan), "streaming kernel: %s", k.C < 48 ? "fewer than 48 chains share too few rows" :
                 L.N % 2 ? "odd sample count (the chain-shared kernels move 16-byte lanes)" : "chosen by BEATAMD_GF_KERNEL or no chain-shared kernel fits this library");
        ctx->gs_ngtp = 0;
    }
    {
        ScopedTimer tm(ctx, standin ? "gfstack_standin" : "gfstack");
        if (k.interp == BEATAMD_NEAREST_NEIGHBOR) {
            if (W == 2) launch_nvar<0, 1, 2>(k.nvar, k.mode, grid, ctx->stream, a);
            else launch_nvar<0, 1, 1>(k.nvar, k.mode, grid, ctx->stream, a);
        } else {
            if (W == 2) launch_nvar<1, 1, 2>(k.nvar, k.mode, grid, ctx->stream, a);
            else launch_nvar<1, 1, 1>(k.nvar, k.mode, grid, ctx->stream, a);
        }
    }
    BA_HIP(hipGetLastError());
    if (k.mode == GF_RESID_SCALAR) BA_TRY(launch_sum_tiles(ctx, a.partial, k.C * L.T, a.ntile, k.quad, standin, 1));
    // stand-in of the runs kernel in mode 3: the misfit of the residuals just stored, only when the tables overflowed
    if (standin && fuse_runs)
        BA_TRY(launch_quadform_banded(ctx, call.band_w, 1, L.N, L.T, k.C, k.out, L.T * L.N, L.N, k.quad, L.T, standin, 1));
    return BEATAMD_OK;
}

}  // namespace beatamd
