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
// Work: C chains x Nobs points x nsrc sources, ~400 flops + 12 transcendentals per evaluation ->
// compute bound on the fp64 VALU; one thread per (chain, observation point).
#include "kernels.hpp"

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
};

struct Vec3 { double x, y, z; };

// Okada (1985) eqs (25)-(30) for one corner term f(xi, eta); returns the bracket contents of the
// strike (ss), dip (ds) and tensile (tf) components
__device__ __forceinline__ void okada_corner(double xi, double eta, double q, double sd, double cd,
                                             double a /* mu/(lambda+mu) */, Vec3 &ss, Vec3 &ds,
                                             Vec3 &tf)
{
    const double R = sqrt(xi * xi + eta * eta + q * q);
    const double yt = eta * cd + q * sd;
    const double dt = eta * sd - q * cd;
    const double X = sqrt(xi * xi + q * q);
    double I1, I2, I3, I4, I5;
    const double lnRe = log(R + eta);
    if (fabs(cd) > 1e-12) {
        I5 = (fabs(xi) < 1e-12)
                 ? 0.0
                 : a * 2.0 / cd * atan((eta * (X + q * cd) + X * (R + X) * sd) / (xi * (R + X) * cd));
        I4 = a / cd * (log(R + dt) - sd * lnRe);
        I3 = a * (yt / (cd * (R + dt)) - lnRe) + sd / cd * I4;
        I1 = a * (-xi / (cd * (R + dt))) - sd / cd * I5;
    } else {
        const double Rd = R + dt;
        I5 = -a * xi * sd / Rd;
        I4 = -a * q / Rd;
        I3 = a / 2.0 * (eta / Rd + yt * q / (Rd * Rd) - lnRe);
        I1 = -a / 2.0 * xi * q / (Rd * Rd);
    }
    I2 = a * (-lnRe) - I3;
    const double at = (fabs(q) < 1e-12) ? 0.0 : atan(xi * eta / (q * R));
    const double Re = R * (R + eta), Rx = R * (R + xi);
    ss.x = xi * q / Re + at + I1 * sd;
    ss.y = yt * q / Re + q * cd / (R + eta) + I2 * sd;
    ss.z = dt * q / Re + q * sd / (R + eta) + I4 * sd;
    ds.x = q / R - I3 * sd * cd;
    ds.y = yt * q / Rx + cd * at - I1 * sd * cd;
    ds.z = dt * q / Rx + sd * at - I5 * sd * cd;
    tf.x = q * q / Re - I3 * sd * sd;
    tf.y = -dt * q / Rx - sd * (xi * q / Re - at) - I1 * sd * sd;
    tf.z = yt * q / Rx + cd * (xi * q / Re - at) - I5 * sd * sd;
}

__device__ __forceinline__ double src_param(const GeomSrcArgs &a, const double *q, int s, int k)
{
    const int64_t o = a.poff[s * GEO_NP + k];
    return o >= 0 ? q[o] : a.pfix[s * GEO_NP + k];
}

// displacement (east, north, up) [m] of source s at the observation point (e, n) [km]
__device__ __forceinline__ void source_disp(const GeomSrcArgs &a, const double *q, int s, double e,
                                            double n, double &ue, double &un, double &uz)
{
    const double D2R = 0.017453292519943295;
    const double es = src_param(a, q, s, 0), ns = src_param(a, q, s, 1);
    const double depth = src_param(a, q, s, 2);
    if (a.kind[s] == 1) {
        // Mogi (1958): (1-nu)/pi * dV * (x, y, d) / R^3 ; km -> m ; volume change in slot 8
        const double dV = src_param(a, q, s, 8);
        const double de = (e - es) * 1e3, dn = (n - ns) * 1e3, d = depth * 1e3;
        const double R2 = de * de + dn * dn + d * d;
        const double cf = (1.0 - a.nu) / 3.141592653589793 * dV / (R2 * sqrt(R2));
        ue = cf * de;
        un = cf * dn;
        uz = cf * d;
        return;
    }
    const double strike = src_param(a, q, s, 3) * D2R, dip = src_param(a, q, s, 4) * D2R;
    const double rake = src_param(a, q, s, 5) * D2R;
    const double L = src_param(a, q, s, 6), W = src_param(a, q, s, 7);
    const double slip = src_param(a, q, s, 8), f = src_param(a, q, s, 9);
    const double sd = sin(dip), cd = cos(dip);
    const double ex = sin(strike), nx = cos(strike);  // along strike
    const double ey = nx, ny = -ex;                    // horizontal down-dip direction
    const double dbot = depth + W * sd;
    const double oe = es - 0.5 * L * ex + W * cd * ey;
    const double on = ns - 0.5 * L * nx + W * cd * ny;
    const double de = e - oe, dn = n - on;
    const double x = de * ex + dn * nx;
    const double y = -(de * ey + dn * ny);
    const double p = y * cd + dbot * sd;
    const double qq = y * sd - dbot * cd;
    const double shear = slip * (1.0 - fabs(f));
    const double U1 = shear * cos(rake), U2 = shear * sin(rake), U3 = slip * f;
    Vec3 ss[4], dsv[4], tf[4];
    okada_corner(x, p, qq, sd, cd, 1.0 - 2.0 * a.nu, ss[0], dsv[0], tf[0]);
    okada_corner(x, p - W, qq, sd, cd, 1.0 - 2.0 * a.nu, ss[1], dsv[1], tf[1]);
    okada_corner(x - L, p, qq, sd, cd, 1.0 - 2.0 * a.nu, ss[2], dsv[2], tf[2]);
    okada_corner(x - L, p - W, qq, sd, cd, 1.0 - 2.0 * a.nu, ss[3], dsv[3], tf[3]);
    const double c2 = 1.0 / (2.0 * 3.141592653589793);
#define CH(V, F) (V[0].F - V[1].F - V[2].F + V[3].F)
    const double ux = -U1 * c2 * CH(ss, x) - U2 * c2 * CH(dsv, x) + U3 * c2 * CH(tf, x);
    const double uy = -U1 * c2 * CH(ss, y) - U2 * c2 * CH(dsv, y) + U3 * c2 * CH(tf, y);
    const double uzz = -U1 * c2 * CH(ss, z) - U2 * c2 * CH(dsv, z) + U3 * c2 * CH(tf, z);
#undef CH
    ue = ux * ex - uy * ey;
    un = ux * nx - uy * ny;
    uz = uzz;
}

__global__ void __launch_bounds__(256) k_geom_los(GeomSrcArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.C * a.Nobs) return;
    const int64_t c = i / a.Nobs, k = i - c * a.Nobs;
    const double *q = a.Q + c * a.nparams;
    const double e = a.east[k], n = a.north[k];
    double ue = 0.0, un = 0.0, uz = 0.0;
    for (int s = 0; s < a.nsrc; s++) {
        double se, sn, sz;
        source_disp(a, q, s, e, n, se, sn, sz);
        ue += se;
        un += sn;
        uz += sz;
    }
    // geodetic.py:642: los_disp = (disp * los_vectors).sum(axis=1) with disp = [n, e, up]
    // (heart.py:4220-4224) and los = [Sn, Se, Su] (heart.py:1381-1410)
    const double *l = a.los + k * 3;
    a.mu[i] = (un * l[0] + ue * l[1]) + uz * l[2];
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
    source_disp(a, a.Q + c * a.nparams, s, a.east[k], a.north[k], ue, un, uz);
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
    const int64_t n = C * nsrc * nobs;
    hipLaunchKernelGGL(k_geom_disp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, out);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

int launch_geom_los(beatamd_ctx *ctx, const GeomSources &g, const double *Q, int64_t nparams,
                    int64_t C, double *mu)
{
    if (C == 0) return BEATAMD_OK;
    GeomSrcArgs a;
    a.nsrc = g.nsrc; a.kind = g.kind; a.poff = g.poff; a.pfix = g.pfix;
    a.Q = Q; a.nparams = nparams; a.C = C; a.Nobs = g.Nobs;
    a.east = g.east; a.north = g.north; a.los = g.los; a.nu = g.nu; a.mu = mu;
    const int64_t n = C * g.Nobs;
    ScopedTimer tm(ctx, "geomlos");
    hipLaunchKernelGGL(k_geom_los, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    BA_HIP(hipGetLastError());
    return BEATAMD_OK;
}

}  // namespace beatamd
