// geometry.hip -- analytic half-space forward models for the geometry-mode geodetic composite
// (BASELINE configs 1 and 2), batched over chains: rectangular dislocation (Okada 1985, surface
// displacements, eqs 25-30) and Mogi point source, projected on the line of sight.
//
// Reference seam: GeodeticGeometryComposite.get_formula (beat/models/geodetic.py:605-659):
//   disp = get_synths(input_rvs)            -> heart.geo_synthetics -> pyrocko engine (NOT in tree)
//   los_disp = (disp * los_vectors).sum(1)  geodetic.py:642
//   residuals = (data - los_disp) * odws ; multivariate_normal_chol
// The displacement arithmetic of the reference lives in pyrocko's GF-store engine (layered
// medium): parity with BEAT is unpinned there.  This kernel is the homogeneous
// half-space counterpart, pinned to the published check values of Okada (1985) Table 2.
//
// Work: C chains x Nobs points x nsrc sources, four corner terms (two arc tangents, two square roots,
// one division, ~60 multiply-adds each) + two logarithms per evaluation -> bound by the fp64 VALU;
// one thread per (chain, observation point), the per-(chain, source) trigonometry shared through LDS.
#include "kernels.hpp"

// multiply-adds may contract here (the Makefile turns contraction off for the kernels that keep the
// reference's operation order; this arithmetic is pinned by tolerance to Okada's published values)
#pragma clang fp contract(fast)

namespace beatamd {

constexpr int GEO_NP = 10;  // east_shift north_shift depth strike dip rake length width slip opening

struct GeomSrcArgs {
    int nsrc;
    const int32_t *kind;     // [nsrc] 0 rectangular, 1 mogi
    const int64_t *poff;     // [nsrc*GEO_NP] offset in q, or -1 -> fixed value
    const double *pfix;      // [nsrc*GEO_NP]
    const double *Q;
    int64_t nparams, C, Nobs;
    const double *east, *north, *los;  // [Nobs], [Nobs], [Nobs,3] (Sn, Se, Su)
    double nu;
    double *mu;  // [C, Nobs] line-of-sight synthetics
    // with res: the weighted residual (data - mu) * odw is stored instead of mu
    const double *data, *odw;
    double *res;
};

struct Vec3 { double x, y, z; };

// Okada (1985) eqs (25)-(30).  The displacement is Chinnery's sum f(x,p) - f(x,p-W) - f(x-L,p) +
// f(x-L,p-W) of a corner term f(xi, eta) that is LINEAR in ln(R+eta) and ln(R+d~) with coefficients
// that depend on the source only.  This kernel is bound by the fp64 VALU (an f64 log / atan / division
// is ~100 / ~100 / ~30 instructions), so the corner terms are accumulated with their signs in a form
// that needs per observation point
//     2 logarithms   ln[(R0+eta0)(R3+eta3) / ((R1+eta1)(R2+eta2))]  and the same for R+d~   (not 8)
//     1 division per corner: 1/R, 1/(R+eta), 1/(R+d~), 1/(R+xi) and the denominator of the I5 arc
//       tangent come out of ONE reciprocal of their product (not 6 divisions)
//     2 arc tangents and 2 square roots per corner (as written in the paper).
struct CornerSum {
    // signed sums of the logarithm-free parts
    double ssx, ssy, ssz, dsx, dsy, dsz, tfx, tfy, tfz;
    double i1, i5, t3, i4r;
    // numerators / denominators of the two logarithm arguments
    double re_n, re_d, rd_n, rd_d;
};

__device__ __forceinline__ void okada_corner(bool PLUS, double xi, double eta, double q, double rq, double sd,
                                             double cd, double rcd, bool vertical,
                                             double a /* mu/(lambda+mu) */, CornerSum &o)
{
    const double R = sqrt(xi * xi + eta * eta + q * q);
    const double yt = eta * cd + q * sd;
    const double dt = eta * sd - q * cd;
    const double X = sqrt(xi * xi + q * q);
    const double Re = R + eta, Rd = R + dt, Rx = R + xi;
    const bool xi0 = fabs(xi) < 1e-12;
    const double den5 = (xi0 || vertical) ? 1.0 : xi * (R + X) * cd;
    // five reciprocals from one division
    const double p1 = R * Re, p2 = p1 * Rd, p3 = p2 * Rx, p4 = p3 * den5;
    const double inv = 1.0 / p4;
    const double rden5 = inv * p3;
    const double i3 = inv * den5;          // 1 / (R Re Rd Rx)
    const double rRx = i3 * p2;
    const double i2 = i3 * Rx;             // 1 / (R Re Rd)
    const double rRd = i2 * p1;
    const double i1_ = i2 * Rd;            // 1 / (R Re)
    const double rRe = i1_ * R;
    const double rR = i1_ * Re;
    double I1, I5, t3, i4r;
    if (!vertical) {
        I5 = xi0 ? 0.0 : a * 2.0 * rcd * atan((eta * (X + q * cd) + X * (R + X) * sd) * rden5);
        t3 = yt * rcd * rRd;
        i4r = 0.0;
        I1 = a * (-xi * rcd * rRd) - sd * rcd * I5;
    } else {
        I5 = -a * xi * sd * rRd;
        i4r = -a * q * rRd;
        t3 = eta * rRd + yt * q * (rRd * rRd);
        I1 = -a / 2.0 * xi * q * (rRd * rRd);
    }
    const double at = (fabs(q) < 1e-12) ? 0.0 : atan(xi * eta * rR * rq);
    const double qRe = q * i1_;            // q / (R (R + eta))
    const double qRx = q * rR * rRx;       // q / (R (R + xi))
    const double w = xi * qRe - at;
    const double sg = PLUS ? 1.0 : -1.0;
#define ACC(F, V) o.F = fma(sg, (V), o.F)
    ACC(ssx, xi * qRe + at);
    ACC(ssy, yt * qRe + q * cd * rRe);
    ACC(ssz, dt * qRe + q * sd * rRe);
    ACC(dsx, q * rR);
    ACC(dsy, yt * qRx + cd * at);
    ACC(dsz, dt * qRx + sd * at);
    ACC(tfx, q * qRe);
    ACC(tfy, -dt * qRx - sd * w);
    ACC(tfz, yt * qRx + cd * w);
    ACC(i1, I1);
    ACC(i5, I5);
    ACC(t3, t3);
    ACC(i4r, i4r);
#undef ACC
    o.re_n *= PLUS ? Re : 1.0;
    o.rd_n *= PLUS ? Rd : 1.0;
    o.re_d *= PLUS ? 1.0 : Re;
    o.rd_d *= PLUS ? 1.0 : Rd;
}

__device__ __forceinline__ double src_param(const GeomSrcArgs &a, const double *q, int s, int k)
{
    const int64_t o = a.poff[s * GEO_NP + k];
    return o >= 0 ? q[o] : a.pfix[s * GEO_NP + k];
}

// what one source of one chain contributes to every observation point: computed once per (chain,
// source) -- six sin/cos and the slip decomposition -- and kept in LDS by k_geom_los
struct SrcConst {
    double oe, on;        // rectangular: Okada origin (east, north) [km] ; Mogi: source position
    double ex, nx;        // along-strike unit vector (east, north)
    double sd, cd, rcd;   // sin / cos / 1/cos of the dip
    double dbot, L, W;    // depth of the lower edge, length, width [km] ; Mogi: depth in dbot
    double U1, U2, U3;    // strike-slip, dip-slip, tensile components ; Mogi: volume change in U1
    int kind;
};

__device__ __forceinline__ void source_const(const GeomSrcArgs &a, const double *q, int s, SrcConst &k)
{
    const double D2R = 0.017453292519943295;
    const double es = src_param(a, q, s, 0), ns = src_param(a, q, s, 1);
    const double depth = src_param(a, q, s, 2);
    k.kind = a.kind[s];
    if (k.kind == 1) {
        k.oe = es; k.on = ns; k.dbot = depth; k.U1 = src_param(a, q, s, 8);
        return;
    }
    const double strike = src_param(a, q, s, 3) * D2R, dip = src_param(a, q, s, 4) * D2R;
    const double rake = src_param(a, q, s, 5) * D2R;
    k.L = src_param(a, q, s, 6);
    k.W = src_param(a, q, s, 7);
    const double slip = src_param(a, q, s, 8), f = src_param(a, q, s, 9);
    k.sd = sin(dip);
    k.cd = cos(dip);
    k.rcd = 1.0 / k.cd;
    k.ex = sin(strike);
    k.nx = cos(strike);
    const double ey = k.nx, ny = -k.ex;                 // horizontal down-dip direction
    k.dbot = depth + k.W * k.sd;
    k.oe = es - 0.5 * k.L * k.ex + k.W * k.cd * ey;
    k.on = ns - 0.5 * k.L * k.nx + k.W * k.cd * ny;
    const double shear = slip * (1.0 - fabs(f));
    k.U1 = shear * cos(rake);
    k.U2 = shear * sin(rake);
    k.U3 = slip * f;
}

// displacement (east, north, up) [m] of one source at the observation point (e, n) [km]
__device__ __forceinline__ void source_disp(const SrcConst &k, double nu, double e, double n, double &ue,
                                            double &un, double &uz)
{
    if (k.kind == 1) {
        // Mogi (1958): (1-nu)/pi * dV * (x, y, d) / R^3 ; km -> m ; volume change in slot 8
        const double de = (e - k.oe) * 1e3, dn = (n - k.on) * 1e3, d = k.dbot * 1e3;
        const double R2 = de * de + dn * dn + d * d;
        const double cf = (1.0 - nu) / 3.141592653589793 * k.U1 / (R2 * sqrt(R2));
        ue = cf * de;
        un = cf * dn;
        uz = cf * d;
        return;
    }
    const double ex = k.ex, nx = k.nx, ey = k.nx, ny = -k.ex, sd = k.sd, cd = k.cd;
    const double de = e - k.oe, dn = n - k.on;
    const double x = de * ex + dn * nx;
    const double y = -(de * ey + dn * ny);
    const double p = y * cd + k.dbot * sd;
    const double qq = y * sd - k.dbot * cd;
    const double al = 1.0 - 2.0 * nu;
    const bool vertical = !(fabs(cd) > 1e-12);
    const double rq = 1.0 / qq;
    CornerSum o;
    o.ssx = o.ssy = o.ssz = o.dsx = o.dsy = o.dsz = o.tfx = o.tfy = o.tfz = 0.0;
    o.i1 = o.i5 = o.t3 = o.i4r = 0.0;
    o.re_n = o.re_d = o.rd_n = o.rd_d = 1.0;
    // f(x,p) - f(x,p-W) - f(x-L,p) + f(x-L,p-W); one corner at a time (not unrolled): the register
    // budget of one corner term lets four waves share a SIMD and hide its dependent fp64 chains
#pragma unroll 1
    for (int cn = 0; cn < 4; cn++)
        okada_corner(cn == 0 || cn == 3, (cn & 2) ? x - k.L : x, (cn & 1) ? p - k.W : p, qq, rq, sd, cd, k.rcd,
                     vertical, al, o);
    // Chinnery sums of the logarithms and of I1..I5 (eqs 28-29)
    const double S1 = log(o.re_n / o.re_d);          // sum +- ln(R + eta)
    double I4, I3;
    if (!vertical) {
        const double S2 = log(o.rd_n / o.rd_d);      // sum +- ln(R + d~)
        I4 = al * k.rcd * (S2 - sd * S1);
        I3 = al * (o.t3 - S1) + sd * k.rcd * I4;
    } else {
        I4 = o.i4r;
        I3 = al / 2.0 * (o.t3 - S1);
    }
    const double I2 = al * (-S1) - I3;
    const double I1 = o.i1, I5 = o.i5;
    const double ssx = o.ssx + I1 * sd, ssy = o.ssy + I2 * sd, ssz = o.ssz + I4 * sd;
    const double dsx = o.dsx - I3 * sd * cd, dsy = o.dsy - I1 * sd * cd, dsz = o.dsz - I5 * sd * cd;
    const double tfx = o.tfx - I3 * sd * sd, tfy = o.tfy - I1 * sd * sd, tfz = o.tfz - I5 * sd * sd;
    const double c2 = 1.0 / (2.0 * 3.141592653589793);
    const double ux = -k.U1 * c2 * ssx - k.U2 * c2 * dsx + k.U3 * c2 * tfx;
    const double uy = -k.U1 * c2 * ssy - k.U2 * c2 * dsy + k.U3 * c2 * tfy;
    const double uzz = -k.U1 * c2 * ssz - k.U2 * c2 * dsz + k.U3 * c2 * tfz;
    ue = ux * ex - uy * ey;
    un = ux * nx - uy * ny;
    uz = uzz;
}

// one thread per (chain, observation point), flat index.  A workgroup touches the chains
// c_first .. c_last of its 256 indices: their source constants are computed once (one thread per
// (chain, source)) into LDS.  With `res` the residual of the likelihood is written instead of the
// synthetics: res = (data - mu) * odw (geodetic.py:1072-1074 / 642-650).
constexpr int GL_MAXC = 48;

// SHARED = false: more (chain, source) pairs per workgroup than the LDS table holds (very few observation
// points): every thread computes its own source constants
template <int WAVES, bool SHARED>
__global__ void __launch_bounds__(256, WAVES) k_geom_los(GeomSrcArgs a)
{
    __shared__ SrcConst sc[SHARED ? GL_MAXC : 1];
    const int64_t total = a.C * a.Nobs;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t i = i0 + threadIdx.x;
    const int64_t c_first = i0 / a.Nobs;
    const int64_t c_last = (min(i0 + 255, total - 1)) / a.Nobs;
    const int ncs = (int)(c_last - c_first + 1) * a.nsrc;
    if (SHARED) {
        if ((int)threadIdx.x < ncs) {
            const int cc = threadIdx.x / a.nsrc, s = threadIdx.x - cc * a.nsrc;
            source_const(a, a.Q + (c_first + cc) * a.nparams, s, sc[threadIdx.x]);
        }
        __syncthreads();
    }
    if (i >= total) return;
    const int64_t c = i / a.Nobs, k = i - c * a.Nobs;
    const double e = a.east[k], n = a.north[k];
    double ue = 0.0, un = 0.0, uz = 0.0;
    for (int s = 0; s < a.nsrc; s++) {
        double se, sn, sz;
        if (SHARED) {
            source_disp(sc[(int)(c - c_first) * a.nsrc + s], a.nu, e, n, se, sn, sz);
        } else {
            SrcConst own;
            source_const(a, a.Q + c * a.nparams, s, own);
            source_disp(own, a.nu, e, n, se, sn, sz);
        }
        ue += se;
        un += sn;
        uz += sz;
    }
    // geodetic.py:642: los_disp = (disp * los_vectors).sum(axis=1) with disp = [n, e, up]
    // (heart.py:4220-4224) and los = [Sn, Se, Su] (heart.py:1381-1410)
    const double *l = a.los + k * 3;
    const double mu = (un * l[0] + ue * l[1]) + uz * l[2];
    if (a.res)
        a.res[i] = (a.data[k] - mu) * a.odw[k];
    else
        a.mu[i] = mu;
}

// heart.geo_synthetics (heart.py:4158-4239) for the half-space engine: one (n, e, up) array per
// (parameter set, source, observation point): out[((c*nsrc + s)*Nobs + k)*3 + {0,1,2}]
__global__ void __launch_bounds__(256) k_geom_disp(GeomSrcArgs a, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.C * a.nsrc * a.Nobs) return;
    const int64_t k = i % a.Nobs;
    const int64_t cs = i / a.Nobs;
    const int s = (int)(cs % a.nsrc);
    const int64_t c = cs / a.nsrc;
    double ue, un, uz;
    SrcConst own;
    source_const(a, a.Q + c * a.nparams, s, own);
    source_disp(own, a.nu, a.east[k], a.north[k], ue, un, uz);
    out[i * 3 + 0] = un;
    out[i * 3 + 1] = ue;
    out[i * 3 + 2] = uz;   // up = -down (heart.py:4222)
}

int launch_geom_disp(beatamd_ctx *ctx, int nsrc, const int32_t *kind, const int64_t *poff,
                     const double *params, int64_t C, int64_t nobs, const double *east,
                     const double *north, double nu, double *out)
{
    if (C == 0 || nobs == 0 || nsrc == 0) return BEATAMD_OK;
    GeomSrcArgs a;
    a.nsrc = nsrc; a.kind = kind; a.poff = poff; a.pfix = params;   // every slot comes from `params`
    a.Q = params; a.nparams = (int64_t)nsrc * GEO_NP; a.C = C; a.Nobs = nobs;
    a.east = east; a.north = north; a.los = nullptr; a.nu = nu; a.mu = nullptr;
    a.data = a.odw = nullptr; a.res = nullptr;
    const int64_t n = C * nsrc * nobs;
    hipLaunchKernelGGL(k_geom_disp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, out);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

int launch_geom_los(beatamd_ctx *ctx, const GeomSources &g, const double *Q, int64_t nparams,
                    int64_t C, double *mu, const double *data, const double *odw, double *res)
{
    if (C == 0) return BEATAMD_OK;
    GeomSrcArgs a;
    a.nsrc = g.nsrc; a.kind = g.kind; a.poff = g.poff; a.pfix = g.pfix;
    a.Q = Q; a.nparams = nparams; a.C = C; a.Nobs = g.Nobs;
    a.east = g.east; a.north = g.north; a.los = g.los; a.nu = g.nu; a.mu = mu;
    a.data = data; a.odw = odw; a.res = res;
    const int64_t n = C * g.Nobs;
    ScopedTimer tm(ctx, "geomlos");
    // waves per SIMD the register budget is cut for (the corner terms are long dependent fp64 chains)
    static const int waves = getenv("BEATAMD_GEOM_WAVES") ? atoi(getenv("BEATAMD_GEOM_WAVES")) : 2;
    const dim3 grid((unsigned)((n + 255) / 256));
    // chains a workgroup of 256 (chain, point) pairs can touch, times the sources
    const int64_t ncs_max = ((255 + g.Nobs - 1) / g.Nobs + 1) * g.nsrc;
    static const bool force_own = getenv("BEATAMD_GEOM_OWN") != nullptr;
    if (ncs_max > GL_MAXC || force_own)
        hipLaunchKernelGGL((k_geom_los<2, false>), grid, dim3(256), 0, ctx->stream, a);
    else if (waves >= 4)
        hipLaunchKernelGGL((k_geom_los<4, true>), grid, dim3(256), 0, ctx->stream, a);
    else if (waves == 3)
        hipLaunchKernelGGL((k_geom_los<3, true>), grid, dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((k_geom_los<2, true>), grid, dim3(256), 0, ctx->stream, a);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd
