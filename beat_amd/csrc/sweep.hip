// sweep.hip -- batched eikonal fast sweeping (rupture onset times) for gfx950.
//
// Reference arithmetic: beat/fast_sweeping/fast_sweep_ext.c:65-206 (eq_solve, upwind,
// fast_sweep).  One 64-lane wavefront owns one (chain, subfault) grid held in LDS.
//
// The reference runs 4 sequential Gauss-Seidel sweeps per outer iteration.  A cell (i,j)
// of a sweep reads the already-updated upwind neighbours (i-1,j),(i,j-1) and the
// not-yet-updated downwind neighbours (in sweep order).  Cells on one anti-diagonal
// i'+j' = k of the sweep-ordered grid neither read nor write each other, and all their
// upwind inputs lie on diagonal k-1, so processing diagonals in order k = 0..ni+nj-2 with
// the lanes spread along the diagonal performs EXACTLY the same double operations on the
// same operands as the sequential loop: results are bitwise those of a sequential
// implementation of the same expressions.  (The reference's glibc pow(x,0.5) is replaced
// by the correctly rounded sqrt; times agree to ~1 ulp)
//
// Built with -ffp-contract=off: no FMA contraction, the C expression order is kept.
#include "kernels.hpp"

namespace beatamd {

__device__ __forceinline__ void wave_lds_sync()
{
    // all lanes of a wavefront run in lockstep; wait for the wave's outstanding LDS
    // operations and stop the compiler from moving LDS accesses across this point
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// fast_sweep_ext.c:65-75
__device__ __forceinline__ double eq_solve(double a, double b, double f, double h)
{
    double v;
    if (fabs(a - b) >= f * h) {
        v = (a < b) ? a : b;
        v += f * h;
    } else {
        double dab = a - b;
        v = a + b + sqrt(2.0 * f * f * h * h - dab * dab);
        v /= 2.0;
    }
    return v;
}

// fast_sweep_ext.c:77-118
__device__ __forceinline__ double upwind(const double *t, int i, int j, const double *slow,
                                         double h, int ni, int nj)
{
    int i1 = i - 1, i2 = i + 1, j1 = j - 1, j2 = j + 1;
    if (i1 < 0) i1 = 0;
    if (i2 >= ni) i2 = ni - 1;
    if (j1 < 0) j1 = 0;
    if (j2 >= nj) j2 = nj - 1;
    double a1 = t[i1 * nj + j], a2 = t[i2 * nj + j];
    double b1 = t[i * nj + j1], b2 = t[i * nj + j2];
    double old = t[i * nj + j];
    double uxmin = (a1 < a2) ? a1 : a2;
    double uymin = (b1 < b2) ? b1 : b2;
    double v = eq_solve(uxmin, uymin, slow[i * nj + j], h);
    return (v < old) ? v : old;
}

// lane l receives x of lane l - 1 (DPP wave_shr:1, a VALU move; lane 0: its own x).  Must run with every lane of the
// wavefront enabled: a DPP read from a disabled lane does not deliver.
__device__ __forceinline__ double from_lane_below(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// One sweep of fast_sweep_ext.c:141-196 by anti-diagonals with the UPWIND operands in registers (grids of at most 64
// rows; round 4).  In sweep order lane ip owns row ip and walks it one cell per diagonal: the upwind neighbour in its
// own row, (ip, jp-1), is the value the lane computed on the previous diagonal, the upwind neighbour in the row below,
// (ip-1, jp), is what lane ip-1 computed on the previous diagonal (one DPP move).  The other operands -- the cell's own
// old value, the two DOWNWIND neighbours (diagonals d+1: not written yet in this sweep) and the slowness -- do not depend
// on anything this sweep has written before diagonal d+1, so they are read from LDS one diagonal ahead.  The dependent
// chain of a diagonal is then DPP -> min -> eq_solve -> min instead of LDS write -> LDS read -> ... (the first version:
// sweep_wave below, kept for grids with more than 64 rows).  Same operands, same operations, same order per cell:
// bitwise the same times.
__device__ __forceinline__ void sweep_diag64(double *t, const double *slow, int ni, int nj, double h, bool irev,
                                              bool jrev, int lane)
{
    const bool row = lane < ni;
    const int i = row ? (irev ? ni - 1 - lane : lane) : 0;
    // true row of sweep row ip + 1; beyond the grid the reference clamps to the cell's own row (its old value)
    const int idn = (lane + 1 < ni) ? (irev ? i - 1 : i + 1) : i;
    double *trow = t + i * nj;
    const double *drow = t + idn * nj, *srow = slow + i * nj;
    // (S, S2: the two slowness terms of eq_solve, f*h and 2*f*f*h*h, formed ahead as well: same expressions, same values.
    // Everything is straight-line: reads go to clamped, always valid cells, both arms of eq_solve are evaluated and
    // selected -- a wavefront that runs alone on its SIMD pays for every branch.)
    double O = 0.0, S = 0.0, S2 = 0.0, Di = 0.0, Dj = 0.0, vprev = 0.0;
    auto fetch = [&](int d, double &o, double &fh, double &c2, double &di, double &dj) {
        const int jc = min(max(d - lane, 0), nj - 1);
        const int j = jrev ? nj - 1 - jc : jc;
        const int jd = (jc + 1 < nj) ? (jrev ? j - 1 : j + 1) : j;    // beyond the row: the cell itself (its old value)
        o = trow[j];
        const double f = srow[j];
        fh = f * h;
        c2 = 2.0 * f * f * h * h;
        di = drow[j];
        dj = trow[jd];
    };
    fetch(0, O, S, S2, Di, Dj);
    for (int d = 0; d < ni + nj - 1; d++) {
        const double below = from_lane_below(vprev);       // (ip-1, jp) of this sweep
        double On, Sn, S2n, Din, Djn;
        fetch(d + 1, On, Sn, S2n, Din, Djn);
        const int jp = d - lane;
        const bool act = row && jp >= 0 && jp < nj;
        const double ui = (lane >= 1) ? below : O;
        const double uj = (jp >= 1) ? vprev : O;
        // fast_sweep_ext.c:77-118 upwind(): a1 = t[i-1][j], a2 = t[i+1][j], b1 = t[i][j-1], b2 = t[i][j+1]
        const double a1 = irev ? Di : ui, a2 = irev ? ui : Di;
        const double b1 = jrev ? Dj : uj, b2 = jrev ? uj : Dj;
        const double uxmin = (a1 < a2) ? a1 : a2;
        const double uymin = (b1 < b2) ? b1 : b2;
        // eq_solve (fast_sweep_ext.c:65-75)
        const double dab = uxmin - uymin;
        const double vlin = ((uxmin < uymin) ? uxmin : uymin) + S;
        const double vsq = (uxmin + uymin + sqrt(S2 - dab * dab)) / 2.0;
        double v = (fabs(dab) >= S) ? vlin : vsq;
        v = (v < O) ? v : O;
        if (act) {
            trow[jrev ? nj - 1 - jp : jp] = v;
            vprev = v;
        }
        O = On; S = Sn; S2 = S2n; Di = Din; Dj = Djn;
    }
}

// fast_sweep_ext.c:120-206, one wavefront.  t/told/slow are this wave's LDS arrays.
__device__ void sweep_wave(double *t, double *told, const double *slow, int ni, int nj, double h,
                           int hi, int hj, int lane, bool a_first_version)
{
    const int n = ni * nj;
    for (int k = lane; k < n; k += 64) t[k] = __builtin_inf();
    wave_lds_sync();
    if (lane == 0) t[hi * nj + hj] = 0.0;
    wave_lds_sync();

    double err = 1.0e6;
    int iter = 0;
    while (err > 0.1) {
        for (int k = lane; k < n; k += 64) told[k] = t[k];
        wave_lds_sync();
        for (int sw = 0; sw < 4; sw++) {
            const bool irev = (sw == 1) || (sw == 2);
            const bool jrev = (sw == 2) || (sw == 3);
            if (ni <= 64 && !a_first_version) {
                sweep_diag64(t, slow, ni, nj, h, irev, jrev, lane);
                wave_lds_sync();
                continue;
            }
            for (int d = 0; d < ni + nj - 1; d++) {
                for (int base = 0; base < ni; base += 64) {
                    const int ip = base + lane;
                    const int jp = d - ip;
                    if (ip < ni && jp >= 0 && jp < nj) {
                        const int i = irev ? (ni - 1 - ip) : ip;
                        const int j = jrev ? (nj - 1 - jp) : jp;
                        const double v = upwind(t, i, j, slow, h, ni, nj);
                        t[i * nj + j] = v;
                    }
                }
                wave_lds_sync();
            }
        }
        // err = sum (new-old)^2  (fast_sweep_ext.c:199-202)
        double e = 0.0;
        for (int k = lane; k < n; k += 64) {
            double dlt = t[k] - told[k];
            e += dlt * dlt;
        }
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
        err = e;
        if (fabs(err - 0.1) <= 1e-9) {
            // the decision is within reach of the summation order: redo the sum in the
            // reference's sequential order
            double es = 0.0;
            for (int k = 0; k < n; k++) {
                double dlt = t[k] - told[k];
                es += dlt * dlt;
            }
            err = es;
        }
        if (++iter >= 100000) break;  // the updates are monotone: never reached; bounds a hang
    }
}

struct SweepParams {
    int mode;  // 0: explicit slowness + integer hypocentres, 1: from the parameter matrix Q
    int64_t nprob;
    // mode 0
    const double *slow;
    const int32_t *hi, *hj;
    int32_t ni, nj;
    double h;
    // mode 1
    const double *Q;
    int64_t nparams, vel_off, nuc_strike_off, nuc_dip_off, time_off, P;
    const int32_t *sf_ndip, *sf_nstrike, *sf_off;
    const double *sf_h;
    int32_t nsub;
    // common
    double *out;
    int *status;
    int32_t *chain_bad;  // mode 1, nullable: chain flagged when its hypocentre index is off the grid
    int32_t nmax;  // LDS doubles reserved per array per wave
    int32_t first_version;   // BEATAMD_SWEEP_V1=1: the LDS-only diagonal loop also for grids of <= 64 rows (A/B, tests)
};

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_fast_sweep(SweepParams a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t prob = (int64_t)blockIdx.x * WAVES + wave;
    if (prob >= a.nprob) return;
    double *t = smem + (size_t)wave * 3 * a.nmax;
    double *told = t + a.nmax;
    double *slow = told + a.nmax;

    int ni, nj, hi, hj;
    double h, tadd = 0.0;
    double *out;
    if (a.mode == 0) {
        ni = a.ni;
        nj = a.nj;
        h = a.h;
        hi = a.hi[prob];
        hj = a.hj[prob];
        const double *s = a.slow + prob * (int64_t)(ni * nj);
        for (int k = lane; k < ni * nj; k += 64) slow[k] = s[k];
        out = a.out + prob * (int64_t)(ni * nj);
    } else {
        const int64_t c = prob / a.nsub;
        const int sf = (int)(prob - c * a.nsub);
        ni = a.sf_ndip[sf];
        nj = a.sf_nstrike[sf];
        h = a.sf_h[sf];
        const double *q = a.Q + c * a.nparams;
        // utility.py:1542-1558 positions2idxs with cell = patch size, min_pos = 0
        // (ffi/fault.py:866-894): round-half-even -> rint, cast through int16
        const double pd = q[a.nuc_dip_off + sf], ps = q[a.nuc_strike_off + sf];
        hi = (int)(int16_t)(long long)rint((pd - 0.0 - (h / 2.0)) / h);
        hj = (int)(int16_t)(long long)rint((ps - 0.0 - (h / 2.0)) / h);
        tadd = q[a.time_off + sf];
        const double *v = q + a.vel_off + a.sf_off[sf];
        // seismic.py:1264: slowness = 1 / velocities
        for (int k = lane; k < ni * nj; k += 64) slow[k] = 1.0 / v[k];
        out = a.out + c * a.P + a.sf_off[sf];
    }
    if (hi < 0 || hi >= ni || hj < 0 || hj >= nj) {
        // the reference writes outside its array here; we flag and clamp
        if (lane == 0) {
            atomicOr(a.status, ST_BAD_HYPO);
            if (a.mode == 1 && a.chain_bad) a.chain_bad[prob / a.nsub] = 1;
        }
        hi = min(max(hi, 0), ni - 1);
        hj = min(max(hj, 0), nj - 1);
    }
    wave_lds_sync();
    sweep_wave(t, told, slow, ni, nj, h, hi, hj, lane, a.first_version != 0);
    wave_lds_sync();
    // seismic.py:1268: starttimes_tmp += time[index]
    for (int k = lane; k < ni * nj; k += 64) out[k] = (a.mode == 0) ? t[k] : (t[k] + tadd);
}

static int launch_sweep(beatamd_ctx *ctx, SweepParams &p, int nmax_cells)
{
    BA_CHECK(nmax_cells > 0 && nmax_cells <= 6400, BEATAMD_EINVAL,
             "fast sweep: subfault with %d patches exceeds the LDS-resident limit (6400)",
             nmax_cells);
    p.nmax = (nmax_cells + 1) & ~1;
    p.status = ctx->d_status;
    p.first_version = GfKnobs::get(gf_knobs(ctx).sweep_v1, 0) != 0 ? 1 : 0;     // (A/B: the round-3 kernel)
    ScopedTimer tm(ctx, "sweep");
    if (p.nmax <= 1600) {
        const int W = 4;
        size_t lds = (size_t)W * 3 * p.nmax * sizeof(double);
        unsigned grid = (unsigned)((p.nprob + W - 1) / W);
        hipLaunchKernelGGL(k_fast_sweep<4>, dim3(grid), dim3(W * 64), lds, ctx->stream, p);
    } else {
        size_t lds = (size_t)3 * p.nmax * sizeof(double);
        if (lds > 64 * 1024)
            BA_HIP(hipFuncSetAttribute((const void *)k_fast_sweep<1>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fast_sweep<1>, dim3((unsigned)p.nprob), dim3(64), lds, ctx->stream,
                           p);
    }
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

int launch_sweep_explicit(beatamd_ctx *ctx, const double *slow, double h, const int32_t *hi,
                          const int32_t *hj, int ni, int nj, int64_t C, double *out)
{
    SweepParams p;
    memset(&p, 0, sizeof(p));
    p.mode = 0;
    p.nprob = C;
    p.slow = slow;
    p.hi = hi;
    p.hj = hj;
    p.ni = ni;
    p.nj = nj;
    p.h = h;
    p.out = out;
    return launch_sweep(ctx, p, ni * nj);
}

int launch_sweep_model(beatamd_ctx *ctx, const FfiModel &m, const double *Q, int64_t C,
                       double *starttimes0, int32_t *chain_bad)
{
    SweepParams p;
    memset(&p, 0, sizeof(p));
    p.mode = 1;
    p.nprob = C * m.nsub;
    p.Q = Q;
    p.nparams = m.layout.nparams;
    p.vel_off = m.layout.velocities_off;
    p.nuc_strike_off = m.layout.nuc_strike_off;
    p.nuc_dip_off = m.layout.nuc_dip_off;
    p.time_off = m.layout.time_off;
    p.P = m.P;
    p.sf_ndip = m.d_ndip;
    p.sf_nstrike = m.d_nstrike;
    p.sf_off = m.d_patch_off;
    p.sf_h = m.d_patch_size;
    p.nsub = m.nsub;
    p.out = starttimes0;
    p.chain_bad = chain_bad;
    int nmax = 0;
    for (int s = 0; s < m.nsub; s++) nmax = std::max(nmax, m.ndip[s] * m.nstrike[s]);
    return launch_sweep(ctx, p, nmax);
}

}  // namespace beatamd
