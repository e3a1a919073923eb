// capi.cpp -- the extern "C" entry points of include/beat_amd.h (host side of the engine).
#include "kernels.hpp"

using namespace beatamd;

namespace {

int dev_alloc_copy(beatamd_ctx *ctx, const void *src, size_t bytes, void **dst)
{
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, bytes ? bytes : 8);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return BEATAMD_ENOMEM;
    }
    if (src && bytes) {
        e = hipMemcpyAsync(d, src, bytes, hipMemcpyDefault, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(d);
            set_error("hipMemcpy(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            return BEATAMD_EHIP;
        }
    }
    *dst = d;
    return BEATAMD_OK;
}

template <class T>
T *get_obj(std::vector<std::unique_ptr<T>> &v, int32_t id)
{
    if (id < 0 || (size_t)id >= v.size()) return nullptr;
    return v[id].get();
}

template <class T>
int32_t add_obj(std::vector<std::unique_ptr<T>> &v, std::unique_ptr<T> o)
{
    for (size_t i = 0; i < v.size(); i++)
        if (!v[i]) {
            v[i] = std::move(o);
            return (int32_t)i;
        }
    v.push_back(std::move(o));
    return (int32_t)v.size() - 1;
}

#define ENTER(ctx)                                                    \
    BA_CHECK((ctx) != nullptr, BEATAMD_EINVAL, "ctx is NULL");        \
    BA_HIP(hipSetDevice((ctx)->device))

// quad[c,d] = ||W_d x||^2 for one weight set, x(c,d,k) = X[c*xs_c + d*xs_d + k]
int wset_quad(beatamd_ctx *ctx, const WeightSet &w, int64_t C, const double *X, int64_t xs_c,
              int64_t xs_d, double *quad)
{
    if (w.kind == BEATAMD_W_SCALAR)
        return launch_scalar_quad(ctx, C, w.nd, w.M, X, xs_c, xs_d, w.w, quad);
    // banded whitening operators (the reference's "exponential" noise structure gives bidiagonal ones): two products per
    // sample instead of a row of the dense matrix
    if (w.band >= 0 && w.wb && GfKnobs::get(gf_knobs(ctx).qf_band, 1) != 0)
        return launch_quadform_banded(ctx, w.wb, w.band, w.M, w.nd, C, X, xs_c, xs_d, quad, w.nd);
    QuadformCall q;
    q.A = w.w;
    q.a_stride = w.M * w.M;
    q.M = w.M;
    q.nd = w.nd;
    q.C = C;
    q.X = X;
    q.xs_c = xs_c;
    q.xs_d = xs_d;
    q.upper_tri = w.upper_tri;
    q.quad = quad;
    q.q_stride = w.nd;
    return launch_quadform(ctx, q);
}

// what remains after the composites wrote their columns: the `like` sum.  A caller that passes a
// LikeTail does that sum itself (the Metropolis step folds it into its accept kernel)
struct LikeTail {
    LikeGroups grp;
    const int32_t *chain_bad = nullptr;
};

// logp_forw_func on device pointers
int ffi_logp_device(beatamd_ctx *ctx, FfiModel &m, int64_t C, const double *Q, double *LL,
                    LikeTail *tail = nullptr)
{
    const int64_t nllk = m.nllk();
    const int64_t np = m.layout.nparams;
    void *p = nullptr;
    LikeGroups grp;
    int64_t col = 0;

    ChainVec slips[4];
    for (int v = 0; v < m.layout.nvar; v++) slips[v] = ChainVec{Q, np, m.layout.slip_off[v]};

    // chains whose indices leave the library grid / the patch grid: like = NaN (rejected by the
    // Metropolis step) in addition to the status word that the next synchronisation raises
    // (only the seismic index maps and the sweep flag chains: without wavemaps there is nothing to clear)
    int32_t *chain_bad = nullptr;
    if (!m.wavemaps.empty()) {
        BA_TRY(ctx->get_scratch(SL_CHAINBAD, (size_t)C * sizeof(int32_t), &p));
        chain_bad = (int32_t *)p;
        BA_HIP(hipMemsetAsync(chain_bad, 0, (size_t)C * sizeof(int32_t), ctx->stream));
    }

    if (!m.wavemaps.empty()) {
        BA_TRY(ctx->get_scratch(SL_ST0, (size_t)C * m.P * sizeof(double), &p));
        double *st0 = (double *)p;
        BA_TRY(launch_sweep_model(ctx, m, Q, C, st0, chain_bad));
        for (auto &wm : m.wavemaps) {
            WeightSet *ws = get_obj(ctx->wsets, wm.wset);
            BA_CHECK(ws, BEATAMD_EINVAL, "wavemap refers to a destroyed weight set");
            GfStackCall k;
            k.nvar = m.layout.nvar;
            for (int v = 0; v < k.nvar; v++) {
                k.libs[v] = get_obj(ctx->seislibs, wm.libs[v]);
                BA_CHECK(k.libs[v] && k.libs[v]->g, BEATAMD_EINVAL,
                         "wavemap refers to a destroyed / empty GF library");
                k.slips[v] = slips[v];
            }
            k.durations = ChainVec{Q, np, m.layout.durations_off};
            k.st.starttimes0 = st0;
            k.st.Q = Q;
            k.st.nparams = np;
            k.order_key[0] = ChainVec{Q, np, m.layout.nuc_strike_off};   // (scheduling hint of k_gfstack_runs)
            k.order_key[1] = ChainVec{Q, np, m.layout.nuc_dip_off};
            k.st.shift_off = wm.shift_off;
            k.st.nslot = wm.nslot; k.st.tslot = wm.tslot; k.st.slot_shift_off = wm.slot_shift_off;
            k.st.chain_bad = chain_bad;
            k.interp = wm.interp;
            k.f32 = wm.f32;
            k.C = C;
            k.data = wm.data;
            BA_TRY(ctx->get_scratch(SL_QUAD, (size_t)C * wm.T * sizeof(double), &p));
            double *quad = (double *)p;
            if (ws->kind == BEATAMD_W_SCALAR) {
                k.mode = GF_RESID_SCALAR;
                k.wscalar = ws->w;
                k.quad = quad;
                BA_TRY(launch_gfstack(ctx, k));
            } else if (ws->band == 1 && ws->wb && ws->M == wm.N && ws->nd == wm.T &&
                       GfKnobs::get(gf_knobs(ctx).qf_band, 1) != 0) {
                // bidiagonal whitening operators (the "exponential" noise structure): the misfit rides in the stacking
                // kernel where it has the epilogue, else residual store + k_quadform_banded (launch_gfstack decides)
                k.mode = GF_RESID_BAND1;
                k.band_w = ws->wb;
                k.quad = quad;
                BA_TRY(ctx->get_scratch(SL_RESID, (size_t)C * wm.T * wm.N * sizeof(double), &p));
                k.out = (double *)p;
                BA_TRY(launch_gfstack(ctx, k));
            } else {
                k.mode = GF_RESID_STORE;
                BA_TRY(ctx->get_scratch(SL_RESID, (size_t)C * wm.T * wm.N * sizeof(double), &p));
                k.out = (double *)p;
                BA_TRY(launch_gfstack(ctx, k));
                BA_TRY(wset_quad(ctx, *ws, C, k.out, wm.T * wm.N, wm.N, quad));
            }
            BA_TRY(launch_mvn_finish(ctx, C, wm.T, wm.N, quad, ws->slog, HpSrc{Q, np, wm.hp_off},
                                     LL + col, nllk));
            col += wm.T;
        }
        grp.end[grp.n++] = (int32_t)col;
    }
    if (m.has_geo) {
        Geodetic &g = m.geo;
        BA_TRY(ctx->get_scratch(SL_MU, (size_t)C * g.Nobs * 2 * sizeof(double), &p));
        double *mu = (double *)p, *res = mu + C * g.Nobs;
        if (m.geo_is_geometry) {
            // synthetics, line of sight and weighted residual in one kernel
            BA_TRY(launch_geom_los(ctx, m.geom, Q, np, C, nullptr, g.data, g.odws, res));
        } else {
            // every slip variable's G.T . slips in one launch (geodetic.py:1065-1070 sums them)
            const GeoLib *gls[4] = {nullptr, nullptr, nullptr, nullptr};
            BA_CHECK(m.layout.nvar <= 4, BEATAMD_EINVAL, "geodetic composite: more than 4 slip variables");
            for (int v = 0; v < m.layout.nvar; v++) {
                gls[v] = get_obj(ctx->geolibs, g.libs[v]);
                BA_CHECK(gls[v], BEATAMD_EINVAL, "geodetic composite refers to a destroyed GF library");
            }
            BA_TRY(launch_geo_stack(ctx, gls, m.layout.nvar, C, slips, 0, mu));
            BA_TRY(launch_geo_residual(ctx, C, g.Nobs, g.data, g.odws, mu, res));
        }
        // small dense datasets (SAR scenes / GNSS of a few hundred points): every dataset's
        // quadratic form and MVN epilogue in one launch; otherwise per dataset on the 64-row tiles
        QuadformSmallCall qs;
        bool small = g.sizes.size() <= 8;
        int64_t o = 0;
        for (size_t d = 0; d < g.sizes.size(); d++) {
            WeightSet *ws = get_obj(ctx->wsets, g.wsets[d]);
            BA_CHECK(ws && ws->nd == 1 && ws->M == g.sizes[d], BEATAMD_EINVAL,
                     "geodetic dataset %zu: weight set missing or of the wrong size", d);
            small = small && ws->kind != BEATAMD_W_SCALAR;
            if (small) {
                qs.A[d] = ws->w; qs.M[d] = ws->M; qs.xoff[d] = o; qs.upper_tri[d] = ws->upper_tri;
                qs.slog[d] = ws->slog; qs.hp_off[d] = g.hp_off + d;
            }
            o += g.sizes[d];
        }
        small = small && quadform_small_applicable((int)g.sizes.size(), qs.M);
        if (small) {
            qs.nd = (int)g.sizes.size();
            qs.C = C; qs.X = res; qs.xs_c = g.Nobs; qs.Q = Q; qs.nparams = np;
            qs.LL = LL + col; qs.ld = nllk;
            BA_TRY(launch_quadform_small(ctx, qs));
        } else {
            BA_TRY(ctx->get_scratch(SL_QUAD, (size_t)C * sizeof(double), &p));
            double *quad = (double *)p;
            o = 0;
            for (size_t d = 0; d < g.sizes.size(); d++) {
                WeightSet *ws = get_obj(ctx->wsets, g.wsets[d]);
                BA_TRY(wset_quad(ctx, *ws, C, res + o, g.Nobs, 0, quad));
                BA_TRY(launch_mvn_finish(ctx, C, 1, ws->M, quad, ws->slog,
                                         HpSrc{Q, np, g.hp_off + d}, LL + col + (int64_t)d, nllk));
                o += g.sizes[d];
            }
        }
        col += (int64_t)g.sizes.size();
        grp.end[grp.n++] = (int32_t)col;
    }
    if (m.lap >= 0) {
        Laplacian *lp = get_obj(ctx->laps, m.lap);
        BA_CHECK(lp, BEATAMD_EINVAL, "model refers to a destroyed laplacian");
        const int nvar = m.layout.nvar;
        BA_TRY(ctx->get_scratch(SL_SLIPS, (size_t)C * nvar * lp->P * sizeof(double), &p));
        double *sl = (double *)p;
        BA_TRY(launch_gather_slips(ctx, C, nvar, lp->P, slips, sl));
        BA_TRY(ctx->get_scratch(SL_QUAD, (size_t)C * nvar * sizeof(double), &p));
        double *quad = (double *)p;
        QuadformCall q;
        q.A = lp->L; q.a_stride = 0; q.M = lp->P; q.nd = nvar; q.C = C;
        q.X = sl; q.xs_c = nvar * lp->P; q.xs_d = lp->P;
        q.quad = quad; q.q_stride = nvar;
        BA_TRY(launch_quadform(ctx, q));
        BA_TRY(launch_laplacian_finish(ctx, C, nvar, lp->P, lp->logdet, quad,
                                       HpSrc{Q + m.layout.h_laplacian_off, np, nullptr}, LL + col,
                                       nllk));
        col += 1;
        grp.end[grp.n++] = (int32_t)col;
    }
    BA_CHECK(col == nllk - 1, BEATAMD_EINVAL, "internal: llk layout mismatch");
    if (tail) {
        tail->grp = grp;
        tail->chain_bad = chain_bad;
        return BEATAMD_OK;
    }
    return launch_like_sum(ctx, C, nllk, grp, LL, chain_bad);
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ fast sweep
int beatamd_fast_sweep_batch(beatamd_ctx *ctx, const double *slowness, double patch_size,
                             const int32_t *h_strk, const int32_t *h_dip, int32_t num_strk,
                             int32_t num_dip, int64_t C, double *out)
{
    ENTER(ctx);
    BA_CHECK(slowness && h_strk && h_dip && out, BEATAMD_EINVAL, "fast_sweep: NULL array");
    BA_CHECK(num_strk > 0 && num_dip > 0 && C >= 0, BEATAMD_EINVAL, "fast_sweep: bad dimensions");
    if (C == 0) return BEATAMD_OK;
    const int64_t n = (int64_t)num_strk * num_dip;
    const void *d_s, *d_hi, *d_hj;
    void *d_o;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, slowness, (size_t)C * n * 8, &d_s));
    BA_TRY(stage_in(ctx, SL_IN1, h_strk, (size_t)C * 4, &d_hi));
    BA_TRY(stage_in(ctx, SL_IN2, h_dip, (size_t)C * 4, &d_hj));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)C * n * 8, &d_o, &rec));
    BA_TRY(launch_sweep_explicit(ctx, (const double *)d_s, patch_size, (const int32_t *)d_hi,
                                 (const int32_t *)d_hj, num_strk, num_dip, C, (double *)d_o));
    return finish_out(ctx, &rec, 1);
}

// ------------------------------------------------------------------ GF libraries
int beatamd_seis_gflib_create(beatamd_ctx *ctx, int64_t T, int64_t P, int64_t D, int64_t S,
                              int64_t N, double st_min, double st_dt, double du_min, double du_dt,
                              int32_t *lib_id)
{
    ENTER(ctx);
    BA_CHECK(lib_id, BEATAMD_EINVAL, "lib_id is NULL");
    BA_CHECK(T > 0 && P > 0 && D > 0 && S > 0 && N > 0, BEATAMD_EINVAL,
             "GF library dimensions must be positive");
    BA_CHECK(st_dt > 0 && du_dt > 0, BEATAMD_EINVAL, "sampling intervals must be positive");
    BA_CHECK(D < 32768 && S < 32768, BEATAMD_EINVAL, "int16 index range exceeded");
    std::unique_ptr<SeisLib> l(new SeisLib());
    l->T = T; l->P = P; l->D = D; l->S = S; l->N = N;
    l->st_min = st_min; l->st_dt = st_dt; l->du_min = du_min; l->du_dt = du_dt;
    *lib_id = add_obj(ctx->seislibs, std::move(l));
    return BEATAMD_OK;
}

static int seis_ensure_storage(beatamd_ctx *ctx, SeisLib *l)
{
    if (l->g) return BEATAMD_OK;
    const size_t bytes = (size_t)l->elems() * sizeof(double);
    hipError_t e = hipMalloc((void **)&l->g, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        l->g = nullptr;
        set_error("GF library of %.2f GB does not fit in HBM: %s", bytes / 1e9,
                  hipGetErrorString(e));
        return BEATAMD_ENOMEM;
    }
    l->owned = true;
    return BEATAMD_OK;
}

// The float copy and the float64 storage hold the same values only as long as nobody rewrites the float64 rows:
// an upload or an in-place (re-)whitening drops the copy and takes the models' wavemaps off it.
static void drop_f32_overlapping(beatamd_ctx *ctx, const void *p, size_t bytes)
{
    const char *a = (const char *)p, *b = a + bytes;
    for (size_t id = 0; id < ctx->seislibs.size(); id++) {
        SeisLib *l = ctx->seislibs[id].get();
        if (!l || !l->g32 || !l->g) continue;
        const char *la = (const char *)l->g, *lb = la + (size_t)l->elems() * sizeof(double);
        if (b <= la || lb <= a) continue;
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(l->g32);
        l->g32 = nullptr;
        for (auto &m : ctx->models) {
            if (!m) continue;
            for (Wavemap &w : m->wavemaps)
                for (int32_t lid : w.libs)
                    if ((size_t)lid == id) w.f32 = false;
        }
    }
}

int beatamd_seis_gflib_round_to_f32(beatamd_ctx *ctx, int32_t lib_id)
{
    ENTER(ctx);
    SeisLib *l = get_obj(ctx->seislibs, lib_id);
    BA_CHECK(l && l->g, BEATAMD_EINVAL, "gflib_round_to_f32: unknown or empty GF library %d", lib_id);
    if (!l->g32) {
        hipError_t e = hipMalloc((void **)&l->g32, (size_t)l->elems() * sizeof(float));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            l->g32 = nullptr;
            set_error("float copy of the GF library (%.2f GB) does not fit in HBM: %s", l->elems() * 4 / 1e9,
                      hipGetErrorString(e));
            return BEATAMD_ENOMEM;
        }
    }
    BA_TRY(launch_round_to_f32(ctx, l->g, l->g32, l->elems()));
    return BEATAMD_OK;
}

int beatamd_ffi_model_set_f32(beatamd_ctx *ctx, int32_t model_id, int32_t wavemap_index, int32_t on)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m && wavemap_index >= 0 && wavemap_index < (int32_t)m->wavemaps.size(), BEATAMD_EINVAL,
             "model_set_f32: unknown model / wavemap");
    Wavemap &w = m->wavemaps[wavemap_index];
    if (on)
        for (int32_t id : w.libs) {
            SeisLib *l = get_obj(ctx->seislibs, id);
            BA_CHECK(l && l->g32, BEATAMD_EINVAL,
                     "model_set_f32: library %d has no float copy (beatamd_seis_gflib_round_to_f32; an upload or a "
                     "re-whitening of its rows drops the copy)", id);
        }
    w.f32 = on != 0;
    return BEATAMD_OK;
}

int beatamd_seis_gflib_upload(beatamd_ctx *ctx, int32_t lib_id, const double *src, int64_t offset,
                              int64_t count)
{
    ENTER(ctx);
    SeisLib *l = get_obj(ctx->seislibs, lib_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown GF library %d", lib_id);
    BA_CHECK(src && offset >= 0 && count >= 0 && offset + count <= l->elems(), BEATAMD_EINVAL,
             "gflib_upload: range [%lld, %lld) outside the library (%lld elements)",
             (long long)offset, (long long)(offset + count), (long long)l->elems());
    BA_TRY(seis_ensure_storage(ctx, l));
    drop_f32_overlapping(ctx, l->g + offset, (size_t)count * 8);
    BA_HIP(hipMemcpyAsync(l->g + offset, src, (size_t)count * 8, hipMemcpyDefault, ctx->stream));
    BA_HIP(hipStreamSynchronize(ctx->stream));
    return BEATAMD_OK;
}

int beatamd_seis_gflib_adopt(beatamd_ctx *ctx, int32_t lib_id, double *device_ptr)
{
    ENTER(ctx);
    SeisLib *l = get_obj(ctx->seislibs, lib_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown GF library %d", lib_id);
    BA_CHECK(device_ptr && is_device_ptr(device_ptr), BEATAMD_EINVAL,
             "gflib_adopt needs a device pointer");
    BA_CHECK(((uintptr_t)device_ptr & 15) == 0, BEATAMD_EINVAL,
             "gflib_adopt: pointer must be 16-byte aligned");
    if (l->owned && l->g) BA_HIP(hipFree(l->g));
    if (l->g32) { BA_HIP(hipFree(l->g32)); l->g32 = nullptr; }   // (a float copy belongs to the old storage)
    l->g = device_ptr;
    l->owned = false;
    return BEATAMD_OK;
}

int beatamd_seis_gflib_device_ptr(beatamd_ctx *ctx, int32_t lib_id, double **device_ptr)
{
    ENTER(ctx);
    SeisLib *l = get_obj(ctx->seislibs, lib_id);
    BA_CHECK(l && device_ptr, BEATAMD_EINVAL, "unknown GF library %d", lib_id);
    BA_TRY(seis_ensure_storage(ctx, l));
    *device_ptr = l->g;
    return BEATAMD_OK;
}

int beatamd_seis_gflib_destroy(beatamd_ctx *ctx, int32_t lib_id)
{
    ENTER(ctx);
    SeisLib *l = get_obj(ctx->seislibs, lib_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown GF library %d", lib_id);
    BA_HIP(hipStreamSynchronize(ctx->stream));
    if (l->owned && l->g) BA_HIP(hipFree(l->g));
    if (l->g32) BA_HIP(hipFree(l->g32));
    ctx->seislibs[lib_id].reset();
    return BEATAMD_OK;
}

int beatamd_seis_stack_all_batch(beatamd_ctx *ctx, int32_t lib_id, int64_t C,
                                 const double *durations, const double *starttimes,
                                 const double *slips, int32_t interpolation, double *out)
{
    ENTER(ctx);
    SeisLib *l = get_obj(ctx->seislibs, lib_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown GF library %d", lib_id);
    BA_CHECK(l->g, BEATAMD_EINVAL, "GF library %d holds no data (upload or adopt first)", lib_id);
    BA_CHECK(durations && starttimes && slips && out, BEATAMD_EINVAL, "stack_all: NULL array");
    BA_CHECK(interpolation == BEATAMD_NEAREST_NEIGHBOR || interpolation == BEATAMD_MULTILINEAR,
             BEATAMD_EINVAL, "Interpolation scheme %d not implemented!", interpolation);
    BA_CHECK(C >= 0, BEATAMD_EINVAL, "negative batch size");
    if (C == 0) return BEATAMD_OK;
    const void *d_du, *d_st, *d_sl;
    void *d_o;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, durations, (size_t)C * l->P * 8, &d_du));
    BA_TRY(stage_in(ctx, SL_IN1, starttimes, (size_t)C * l->T * l->P * 8, &d_st));
    BA_TRY(stage_in(ctx, SL_IN2, slips, (size_t)C * l->P * 8, &d_sl));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)C * l->T * l->N * 8, &d_o, &rec));
    GfStackCall k;
    k.libs[0] = l;
    k.nvar = 1;
    k.slips[0] = ChainVec{(const double *)d_sl, l->P, 0};
    k.durations = ChainVec{(const double *)d_du, l->P, 0};
    k.st.explicit_st = (const double *)d_st;
    k.interp = interpolation;
    k.C = C;
    k.mode = GF_STORE_SYN;
    k.out = (double *)d_o;
    BA_TRY(launch_gfstack(ctx, k));
    return finish_out(ctx, &rec, 1);
}

int beatamd_geo_gflib_create(beatamd_ctx *ctx, int64_t P, int64_t Nobs, const double *G,
                             int32_t *lib_id)
{
    ENTER(ctx);
    BA_CHECK(lib_id && G && P > 0 && Nobs > 0, BEATAMD_EINVAL, "geo_gflib_create: bad argument");
    std::unique_ptr<GeoLib> l(new GeoLib());
    l->P = P;
    l->Nobs = Nobs;
    BA_TRY(dev_alloc_copy(ctx, G, (size_t)P * Nobs * 8, (void **)&l->g));
    *lib_id = add_obj(ctx->geolibs, std::move(l));
    return BEATAMD_OK;
}

int beatamd_geo_gflib_destroy(beatamd_ctx *ctx, int32_t lib_id)
{
    ENTER(ctx);
    GeoLib *l = get_obj(ctx->geolibs, lib_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown geodetic GF library %d", lib_id);
    BA_HIP(hipStreamSynchronize(ctx->stream));
    if (l->g) BA_HIP(hipFree(l->g));
    ctx->geolibs[lib_id].reset();
    return BEATAMD_OK;
}

int beatamd_geo_stack_all_batch(beatamd_ctx *ctx, int32_t lib_id, int64_t C, const double *slips,
                                int32_t accumulate, double *out)
{
    ENTER(ctx);
    GeoLib *l = get_obj(ctx->geolibs, lib_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown geodetic GF library %d", lib_id);
    BA_CHECK(slips && out && C >= 0, BEATAMD_EINVAL, "geo_stack_all: bad argument");
    if (C == 0) return BEATAMD_OK;
    const void *d_sl;
    void *d_o;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, slips, (size_t)C * l->P * 8, &d_sl));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)C * l->Nobs * 8, &d_o, &rec, accumulate != 0));
    const GeoLib *one[1] = {l};
    const ChainVec sv[1] = {ChainVec{(const double *)d_sl, l->P, 0}};
    BA_TRY(launch_geo_stack(ctx, one, 1, C, sv, accumulate, (double *)d_o));
    return finish_out(ctx, &rec, 1);
}

// ------------------------------------------------------------------ likelihood
static int wset_fill(beatamd_ctx *ctx, WeightSet *w, const double *weights, const double *slog)
{
    const size_t wbytes =
        (size_t)(w->kind == BEATAMD_W_SCALAR ? w->nd : w->nd * w->M * w->M) * sizeof(double);
    BA_HIP(hipMemcpyAsync(w->w, weights, wbytes, hipMemcpyDefault, ctx->stream));
    BA_HIP(hipMemcpyAsync(w->slog, slog, (size_t)w->nd * 8, hipMemcpyDefault, ctx->stream));
    BA_HIP(hipStreamSynchronize(ctx->stream));
    w->upper_tri = 0;
    if (w->kind == BEATAMD_W_DENSE) {
        void *p = nullptr;
        BA_TRY(ctx->get_scratch(SL_MISC, 64, &p));
        BA_TRY(launch_check_upper_tri(ctx, w->w, w->nd, w->M, (int *)p));
        int flag = 0;
        BA_HIP(hipMemcpyAsync(&flag, p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        BA_HIP(hipStreamSynchronize(ctx->stream));
        w->upper_tri = flag;
        // banded?  (upper-triangular with nothing beyond a few columns right of the diagonal)
        w->band = -1;
        if (w->wb) { BA_HIP(hipFree(w->wb)); w->wb = nullptr; }
        if (flag && w->M > 2 * QF_BAND_LIMIT) {
            BA_TRY(ctx->get_scratch(SL_MISC, (size_t)w->nd * w->M * 8 + 64, &p));
            int64_t band = -1;
            w->dropped_rel = 0.0;
            BA_TRY(launch_band_detect(ctx, w->w, w->nd, w->M, p, &band, &w->dropped_rel));
            // the choice is visible: beatamd_weights_band_info, and one line under BEATAMD_VERBOSE
            if (getenv("BEATAMD_VERBOSE"))
                fprintf(stderr, "beat_amd: weight set of %lld operators of %lld^2: %s (half bandwidth %lld, largest entry beyond it "
                                "%.3g of its row's largest)\n", (long long)w->nd, (long long)w->M,
                        band <= QF_BAND_LIMIT ? "evaluated on its band" : "dense", (long long)band, w->dropped_rel);
            if (band >= 0 && band <= QF_BAND_LIMIT) {
                BA_TRY(dev_alloc_copy(ctx, nullptr, (size_t)(w->nd * w->M * (band + 1)) * sizeof(double), (void **)&w->wb));
                BA_TRY(launch_band_pack(ctx, w->w, w->nd, w->M, band, w->wb));
                w->band = band;
            }
        }
    }
    return BEATAMD_OK;
}

int beatamd_weights_create(beatamd_ctx *ctx, int32_t kind, int64_t nd, int64_t M,
                           const double *weights, const double *slog_pdet, int32_t *wset_id)
{
    ENTER(ctx);
    BA_CHECK(kind == BEATAMD_W_SCALAR || kind == BEATAMD_W_DENSE, BEATAMD_EINVAL,
             "unknown weight kind %d", kind);
    BA_CHECK(nd > 0 && M > 0 && weights && slog_pdet && wset_id, BEATAMD_EINVAL,
             "weights_create: bad argument");
    std::unique_ptr<WeightSet> w(new WeightSet());
    w->kind = kind;
    w->nd = nd;
    w->M = M;
    const size_t wbytes = (size_t)(kind == BEATAMD_W_SCALAR ? nd : nd * M * M) * sizeof(double);
    BA_TRY(dev_alloc_copy(ctx, nullptr, wbytes, (void **)&w->w));
    BA_TRY(dev_alloc_copy(ctx, nullptr, (size_t)nd * 8, (void **)&w->slog));
    BA_TRY(wset_fill(ctx, w.get(), weights, slog_pdet));
    *wset_id = add_obj(ctx->wsets, std::move(w));
    return BEATAMD_OK;
}

int beatamd_weights_update(beatamd_ctx *ctx, int32_t wset_id, int32_t kind, int64_t count,
                           const double *weights, const double *slog_pdet)
{
    ENTER(ctx);
    WeightSet *w = get_obj(ctx->wsets, wset_id);
    BA_CHECK(w && weights && slog_pdet, BEATAMD_EINVAL, "weights_update: bad argument");
    BA_CHECK(kind == w->kind, BEATAMD_EINVAL,
             "weights_update: weight set %d holds %s weights, got %s ones (a pre-whitened wavemap "
             "keeps scalar weights: re-whiten the library instead)", wset_id,
             w->kind == BEATAMD_W_SCALAR ? "scalar" : "dense", kind == BEATAMD_W_SCALAR ? "scalar" : "dense");
    const int64_t want = w->kind == BEATAMD_W_SCALAR ? w->nd : w->nd * w->M * w->M;
    BA_CHECK(count == want, BEATAMD_EINVAL, "weights_update: %lld elements given, the set holds %lld",
             (long long)count, (long long)want);
    BA_HIP(hipStreamSynchronize(ctx->stream));
    return wset_fill(ctx, w, weights, slog_pdet);
}

int beatamd_weights_band_info(beatamd_ctx *ctx, int32_t wset_id, int64_t *band, double *max_dropped_rel)
{
    ENTER(ctx);
    WeightSet *w = get_obj(ctx->wsets, wset_id);
    BA_CHECK(w && band && max_dropped_rel, BEATAMD_EINVAL, "weights_band_info: bad argument");
    const bool on = w->kind == BEATAMD_W_DENSE && w->wb && GfKnobs::get(gf_knobs(ctx).qf_band, 1) != 0;
    *band = on ? w->band : -1;
    *max_dropped_rel = on ? w->dropped_rel : 0.0;
    return BEATAMD_OK;
}

int beatamd_weights_band(beatamd_ctx *ctx, int32_t wset_id, int64_t *band)
{
    ENTER(ctx);
    WeightSet *w = get_obj(ctx->wsets, wset_id);
    BA_CHECK(w && band, BEATAMD_EINVAL, "weights_band: unknown weight set %d", wset_id);
    *band = (w->kind == BEATAMD_W_DENSE && w->wb && GfKnobs::get(gf_knobs(ctx).qf_band, 1) != 0) ? w->band : -1;
    return BEATAMD_OK;
}

int beatamd_weights_destroy(beatamd_ctx *ctx, int32_t wset_id)
{
    ENTER(ctx);
    WeightSet *w = get_obj(ctx->wsets, wset_id);
    BA_CHECK(w, BEATAMD_EINVAL, "unknown weight set %d", wset_id);
    BA_HIP(hipStreamSynchronize(ctx->stream));
    if (w->w) BA_HIP(hipFree(w->w));
    if (w->wb) BA_HIP(hipFree(w->wb));
    if (w->slog) BA_HIP(hipFree(w->slog));
    ctx->wsets[wset_id].reset();
    return BEATAMD_OK;
}

int beatamd_mvn_chol_logp_batch(beatamd_ctx *ctx, int32_t wset_id, int64_t C,
                                const double *residuals, const double *hp, double *logpts)
{
    ENTER(ctx);
    WeightSet *w = get_obj(ctx->wsets, wset_id);
    BA_CHECK(w, BEATAMD_EINVAL, "unknown weight set %d", wset_id);
    BA_CHECK(residuals && hp && logpts && C >= 0, BEATAMD_EINVAL, "mvn_chol_logp: bad argument");
    if (C == 0) return BEATAMD_OK;
    const void *d_r, *d_h;
    void *d_o, *p;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, residuals, (size_t)C * w->nd * w->M * 8, &d_r));
    BA_TRY(stage_in(ctx, SL_IN1, hp, (size_t)C * w->nd * 8, &d_h));
    BA_TRY(stage_out(ctx, SL_OUT0, logpts, (size_t)C * w->nd * 8, &d_o, &rec));
    BA_TRY(ctx->get_scratch(SL_QUAD, (size_t)C * w->nd * 8, &p));
    BA_TRY(wset_quad(ctx, *w, C, (const double *)d_r, w->nd * w->M, w->M, (double *)p));
    BA_TRY(launch_mvn_finish(ctx, C, w->nd, w->M, (const double *)p, w->slog,
                             HpSrc{(const double *)d_h, w->nd, nullptr}, (double *)d_o, w->nd));
    return finish_out(ctx, &rec, 1);
}

int beatamd_laplacian_create(beatamd_ctx *ctx, int64_t P, const double *L, double logdet,
                             int32_t *lap_id)
{
    ENTER(ctx);
    BA_CHECK(P > 0 && L && lap_id, BEATAMD_EINVAL, "laplacian_create: bad argument");
    std::unique_ptr<Laplacian> l(new Laplacian());
    l->P = P;
    l->logdet = logdet;
    BA_TRY(dev_alloc_copy(ctx, L, (size_t)P * P * 8, (void **)&l->L));
    *lap_id = add_obj(ctx->laps, std::move(l));
    return BEATAMD_OK;
}

int beatamd_laplacian_destroy(beatamd_ctx *ctx, int32_t lap_id)
{
    ENTER(ctx);
    Laplacian *l = get_obj(ctx->laps, lap_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown laplacian %d", lap_id);
    BA_HIP(hipStreamSynchronize(ctx->stream));
    if (l->L) BA_HIP(hipFree(l->L));
    ctx->laps[lap_id].reset();
    return BEATAMD_OK;
}

int beatamd_laplacian_logp_batch(beatamd_ctx *ctx, int32_t lap_id, int64_t C, int64_t nvar,
                                 const double *slips, const double *hp, double *out)
{
    ENTER(ctx);
    Laplacian *l = get_obj(ctx->laps, lap_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown laplacian %d", lap_id);
    BA_CHECK(slips && hp && out && C >= 0 && nvar > 0, BEATAMD_EINVAL, "laplacian_logp: bad argument");
    if (C == 0) return BEATAMD_OK;
    const void *d_s, *d_h;
    void *d_o, *p;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, slips, (size_t)C * nvar * l->P * 8, &d_s));
    BA_TRY(stage_in(ctx, SL_IN1, hp, (size_t)C * 8, &d_h));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)C * 8, &d_o, &rec));
    BA_TRY(ctx->get_scratch(SL_QUAD, (size_t)C * nvar * 8, &p));
    QuadformCall q;
    q.A = l->L; q.a_stride = 0; q.M = l->P; q.nd = nvar; q.C = C;
    q.X = (const double *)d_s; q.xs_c = nvar * l->P; q.xs_d = l->P;
    q.quad = (double *)p; q.q_stride = nvar;
    BA_TRY(launch_quadform(ctx, q));
    BA_TRY(launch_laplacian_finish(ctx, C, nvar, l->P, l->logdet, (const double *)p,
                                   HpSrc{(const double *)d_h, 1, nullptr}, (double *)d_o, 1));
    return finish_out(ctx, &rec, 1);
}

// ------------------------------------------------------------------ fused FFI model
int beatamd_ffi_model_create(beatamd_ctx *ctx, const beatamd_ffi_layout *layout, int32_t nsub,
                             const int32_t *ndip, const int32_t *nstrike, const double *patch_size,
                             int32_t *model_id)
{
    ENTER(ctx);
    BA_CHECK(layout && model_id, BEATAMD_EINVAL, "ffi_model_create: bad argument");
    BA_CHECK(layout->nvar >= 0 && layout->nvar <= 3, BEATAMD_EINVAL,
             "0..3 slip variables supported, got %d", layout->nvar);
    BA_CHECK(nsub >= 0 && (nsub == 0 || (ndip && nstrike && patch_size)), BEATAMD_EINVAL,
             "ffi_model_create: subfault description missing");
    std::unique_ptr<FfiModel> m(new FfiModel());
    m->layout = *layout;
    m->nsub = nsub;
    int64_t P = 0;
    for (int s = 0; s < nsub; s++) {
        BA_CHECK(ndip[s] > 0 && nstrike[s] > 0 && patch_size[s] > 0, BEATAMD_EINVAL,
                 "subfault %d: bad discretisation", s);
        m->ndip.push_back(ndip[s]);
        m->nstrike.push_back(nstrike[s]);
        m->patch_size.push_back(patch_size[s]);
        m->patch_off.push_back((int32_t)P);
        P += (int64_t)ndip[s] * nstrike[s];
    }
    m->P = P;
    if (nsub > 0) {
        BA_TRY(dev_alloc_copy(ctx, m->ndip.data(), nsub * 4, (void **)&m->d_ndip));
        BA_TRY(dev_alloc_copy(ctx, m->nstrike.data(), nsub * 4, (void **)&m->d_nstrike));
        BA_TRY(dev_alloc_copy(ctx, m->patch_off.data(), nsub * 4, (void **)&m->d_patch_off));
        BA_TRY(dev_alloc_copy(ctx, m->patch_size.data(), nsub * 8, (void **)&m->d_patch_size));
    }
    *model_id = add_obj(ctx->models, std::move(m));
    return BEATAMD_OK;
}

int beatamd_ffi_model_add_wavemap(beatamd_ctx *ctx, int32_t model_id, const int32_t *lib_ids,
                                  const double *data, int32_t wset_id, const int64_t *hp_off,
                                  const int64_t *shift_off, int32_t interpolation)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    BA_CHECK(lib_ids && data && hp_off, BEATAMD_EINVAL, "add_wavemap: NULL argument");
    BA_CHECK(m->nsub > 0, BEATAMD_EINVAL, "add_wavemap: model has no subfaults");
    BA_CHECK(interpolation == BEATAMD_NEAREST_NEIGHBOR || interpolation == BEATAMD_MULTILINEAR,
             BEATAMD_EINVAL, "Interpolation scheme %d not implemented!", interpolation);
    Wavemap w;
    SeisLib *l0 = nullptr;
    for (int v = 0; v < m->layout.nvar; v++) {
        SeisLib *l = get_obj(ctx->seislibs, lib_ids[v]);
        BA_CHECK(l, BEATAMD_EINVAL, "add_wavemap: unknown GF library %d", lib_ids[v]);
        if (!l0) l0 = l;
        w.libs.push_back(lib_ids[v]);
    }
    BA_CHECK(l0->P == m->P, BEATAMD_EINVAL,
             "add_wavemap: library has %lld patches, fault has %lld", (long long)l0->P,
             (long long)m->P);
    WeightSet *ws = get_obj(ctx->wsets, wset_id);
    BA_CHECK(ws && ws->nd == l0->T && ws->M == l0->N, BEATAMD_EINVAL,
             "add_wavemap: weight set must hold %lld datasets of %lld samples", (long long)l0->T,
             (long long)l0->N);
    w.T = l0->T;
    w.N = l0->N;
    w.wset = wset_id;
    w.interp = interpolation;
    const int64_t np = m->layout.nparams;
    for (int64_t t = 0; t < w.T; t++) {
        BA_CHECK(hp_off[t] >= 0 && hp_off[t] < np, BEATAMD_EINVAL, "add_wavemap: hp_off[%lld] outside q",
                 (long long)t);
        if (shift_off)
            BA_CHECK(shift_off[t] >= 0 && shift_off[t] < np, BEATAMD_EINVAL,
                     "add_wavemap: shift_off[%lld] outside q", (long long)t);
    }
    BA_TRY(dev_alloc_copy(ctx, data, (size_t)w.T * w.N * 8, (void **)&w.data));
    BA_TRY(dev_alloc_copy(ctx, hp_off, (size_t)w.T * 8, (void **)&w.hp_off));
    if (shift_off) {
        BA_TRY(dev_alloc_copy(ctx, shift_off, (size_t)w.T * 8, (void **)&w.shift_off));
        // slots in the order of first appearance
        std::vector<int32_t> tslot((size_t)w.T);
        std::vector<int64_t> slot_off;
        for (int64_t t = 0; t < w.T; t++) {
            size_t k = 0;
            while (k < slot_off.size() && slot_off[k] != shift_off[t]) k++;
            if (k == slot_off.size()) slot_off.push_back(shift_off[t]);
            tslot[(size_t)t] = (int32_t)k;
        }
        if ((int64_t)slot_off.size() < w.T) {
            w.nslot = (int32_t)slot_off.size();
            BA_TRY(dev_alloc_copy(ctx, tslot.data(), tslot.size() * sizeof(int32_t), (void **)&w.tslot));
            BA_TRY(dev_alloc_copy(ctx, slot_off.data(), slot_off.size() * sizeof(int64_t), (void **)&w.slot_shift_off));
        }
    }
    m->wavemaps.push_back(std::move(w));
    return BEATAMD_OK;
}

int beatamd_ffi_model_add_geodetic(beatamd_ctx *ctx, int32_t model_id, const int32_t *geo_lib_ids,
                                   const double *data, const double *odws, int32_t nd,
                                   const int64_t *sizes, const int32_t *wset_ids,
                                   const int64_t *hp_off)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    BA_CHECK(!m->has_geo, BEATAMD_EINVAL, "model already has a geodetic composite");
    BA_CHECK(geo_lib_ids && data && odws && sizes && wset_ids && hp_off && nd > 0, BEATAMD_EINVAL,
             "add_geodetic: bad argument");
    Geodetic g;
    int64_t nobs = 0;
    for (int d = 0; d < nd; d++) {
        BA_CHECK(sizes[d] > 0, BEATAMD_EINVAL, "add_geodetic: empty dataset %d", d);
        BA_CHECK(hp_off[d] >= 0 && hp_off[d] < m->layout.nparams, BEATAMD_EINVAL,
                 "add_geodetic: hp_off[%d] outside q", d);
        WeightSet *ws = get_obj(ctx->wsets, wset_ids[d]);
        BA_CHECK(ws && ws->nd == 1 && ws->M == sizes[d], BEATAMD_EINVAL,
                 "add_geodetic: weight set of dataset %d must be 1 x %lld", d, (long long)sizes[d]);
        g.sizes.push_back(sizes[d]);
        g.wsets.push_back(wset_ids[d]);
        nobs += sizes[d];
    }
    for (int v = 0; v < m->layout.nvar; v++) {
        GeoLib *l = get_obj(ctx->geolibs, geo_lib_ids[v]);
        BA_CHECK(l && l->Nobs == nobs, BEATAMD_EINVAL,
                 "add_geodetic: library %d missing or observation count mismatch", geo_lib_ids[v]);
        if (m->nsub > 0)
            BA_CHECK(l->P == m->P, BEATAMD_EINVAL, "add_geodetic: patch count mismatch");
        else
            m->P = l->P;
        g.libs.push_back(geo_lib_ids[v]);
    }
    g.Nobs = nobs;
    BA_TRY(dev_alloc_copy(ctx, data, (size_t)nobs * 8, (void **)&g.data));
    BA_TRY(dev_alloc_copy(ctx, odws, (size_t)nobs * 8, (void **)&g.odws));
    BA_TRY(dev_alloc_copy(ctx, hp_off, (size_t)nd * 8, (void **)&g.hp_off));
    m->geo = std::move(g);
    m->has_geo = true;
    return BEATAMD_OK;
}

int beatamd_ffi_model_add_geodetic_geometry(beatamd_ctx *ctx, int32_t model_id, int32_t nsrc,
                                            const int32_t *kind, const int64_t *param_off,
                                            const double *param_fixed, int64_t nobs,
                                            const double *east, const double *north,
                                            const double *los, double nu, const double *data,
                                            const double *odws, int32_t nd, const int64_t *sizes,
                                            const int32_t *wset_ids, const int64_t *hp_off)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    BA_CHECK(!m->has_geo, BEATAMD_EINVAL, "model already has a geodetic composite");
    BA_CHECK(nsrc > 0 && kind && param_off && param_fixed && east && north && los && data && odws &&
                 sizes && wset_ids && hp_off && nd > 0 && nobs > 0,
             BEATAMD_EINVAL, "add_geodetic_geometry: bad argument");
    BA_CHECK(nu > -1.0 && nu < 0.5, BEATAMD_EINVAL, "Poisson ratio outside (-1, 0.5)");
    Geodetic g;
    int64_t tot = 0;
    for (int d = 0; d < nd; d++) {
        BA_CHECK(sizes[d] > 0, BEATAMD_EINVAL, "add_geodetic_geometry: empty dataset %d", d);
        BA_CHECK(hp_off[d] >= 0 && hp_off[d] < m->layout.nparams, BEATAMD_EINVAL,
                 "add_geodetic_geometry: hp_off[%d] outside q", d);
        WeightSet *ws = get_obj(ctx->wsets, wset_ids[d]);
        BA_CHECK(ws && ws->nd == 1 && ws->M == sizes[d], BEATAMD_EINVAL,
                 "add_geodetic_geometry: weight set of dataset %d must be 1 x %lld", d,
                 (long long)sizes[d]);
        g.sizes.push_back(sizes[d]);
        g.wsets.push_back(wset_ids[d]);
        tot += sizes[d];
    }
    BA_CHECK(tot == nobs, BEATAMD_EINVAL, "add_geodetic_geometry: dataset sizes sum to %lld, not %lld",
             (long long)tot, (long long)nobs);
    for (int i = 0; i < nsrc * 10; i++)
        BA_CHECK(param_off[i] < m->layout.nparams, BEATAMD_EINVAL,
                 "add_geodetic_geometry: parameter offset %d outside q", i);
    for (int i = 0; i < nsrc; i++)
        BA_CHECK(kind[i] == 0 || kind[i] == 1, BEATAMD_EINVAL, "unknown source kind %d", kind[i]);
    g.Nobs = nobs;
    BA_TRY(dev_alloc_copy(ctx, data, (size_t)nobs * 8, (void **)&g.data));
    BA_TRY(dev_alloc_copy(ctx, odws, (size_t)nobs * 8, (void **)&g.odws));
    BA_TRY(dev_alloc_copy(ctx, hp_off, (size_t)nd * 8, (void **)&g.hp_off));
    GeomSources &gs = m->geom;
    gs.nsrc = nsrc;
    gs.Nobs = nobs;
    gs.nu = nu;
    BA_TRY(dev_alloc_copy(ctx, kind, (size_t)nsrc * 4, (void **)&gs.kind));
    BA_TRY(dev_alloc_copy(ctx, param_off, (size_t)nsrc * 10 * 8, (void **)&gs.poff));
    BA_TRY(dev_alloc_copy(ctx, param_fixed, (size_t)nsrc * 10 * 8, (void **)&gs.pfix));
    BA_TRY(dev_alloc_copy(ctx, east, (size_t)nobs * 8, (void **)&gs.east));
    BA_TRY(dev_alloc_copy(ctx, north, (size_t)nobs * 8, (void **)&gs.north));
    BA_TRY(dev_alloc_copy(ctx, los, (size_t)nobs * 3 * 8, (void **)&gs.los));
    m->geo = std::move(g);
    m->has_geo = true;
    m->geo_is_geometry = true;
    return BEATAMD_OK;
}

int beatamd_ffi_model_set_laplacian(beatamd_ctx *ctx, int32_t model_id, int32_t lap_id)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    Laplacian *l = get_obj(ctx->laps, lap_id);
    BA_CHECK(l, BEATAMD_EINVAL, "unknown laplacian %d", lap_id);
    BA_CHECK(m->layout.h_laplacian_off >= 0 && m->layout.h_laplacian_off < m->layout.nparams,
             BEATAMD_EINVAL, "set_laplacian: layout has no h_laplacian offset");
    BA_CHECK(m->P == 0 || l->P == m->P, BEATAMD_EINVAL, "set_laplacian: patch count mismatch");
    if (m->P == 0) m->P = l->P;
    m->lap = lap_id;
    return BEATAMD_OK;
}

int beatamd_ffi_model_nllk(beatamd_ctx *ctx, int32_t model_id, int64_t *nllk)
{
    BA_CHECK(ctx && nllk, BEATAMD_EINVAL, "bad argument");
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    *nllk = m->nllk();
    return BEATAMD_OK;
}

int beatamd_ffi_model_destroy(beatamd_ctx *ctx, int32_t model_id)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    BA_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &w : m->wavemaps) {
        if (w.data) (void)hipFree(w.data);
        if (w.hp_off) (void)hipFree(w.hp_off);
        if (w.shift_off) (void)hipFree(w.shift_off);
        if (w.tslot) (void)hipFree(w.tslot);
        if (w.slot_shift_off) (void)hipFree(w.slot_shift_off);
    }
    if (m->geo.data) (void)hipFree(m->geo.data);
    if (m->geo.odws) (void)hipFree(m->geo.odws);
    if (m->geo.hp_off) (void)hipFree(m->geo.hp_off);
    if (m->geom.kind) (void)hipFree(m->geom.kind);
    if (m->geom.poff) (void)hipFree(m->geom.poff);
    if (m->geom.pfix) (void)hipFree(m->geom.pfix);
    if (m->geom.east) (void)hipFree(m->geom.east);
    if (m->geom.north) (void)hipFree(m->geom.north);
    if (m->geom.los) (void)hipFree(m->geom.los);
    if (m->d_ndip) (void)hipFree(m->d_ndip);
    if (m->d_nstrike) (void)hipFree(m->d_nstrike);
    if (m->d_patch_off) (void)hipFree(m->d_patch_off);
    if (m->d_patch_size) (void)hipFree(m->d_patch_size);
    ctx->models[model_id].reset();
    return BEATAMD_OK;
}

static int model_check_layout(const FfiModel &m)
{
    const beatamd_ffi_layout &L = m.layout;
    const int64_t np = L.nparams;
    BA_CHECK(np > 0, BEATAMD_EINVAL, "layout: nparams must be positive");
    for (int v = 0; v < L.nvar; v++)
        BA_CHECK(L.slip_off[v] >= 0 && L.slip_off[v] + m.P <= np, BEATAMD_EINVAL,
                 "layout: slip variable %d outside q", v);
    if (!m.wavemaps.empty()) {
        BA_CHECK(L.durations_off >= 0 && L.durations_off + m.P <= np, BEATAMD_EINVAL,
                 "layout: durations outside q");
        BA_CHECK(L.velocities_off >= 0 && L.velocities_off + m.P <= np, BEATAMD_EINVAL,
                 "layout: velocities outside q");
        BA_CHECK(L.nuc_strike_off >= 0 && L.nuc_strike_off + m.nsub <= np && L.nuc_dip_off >= 0 &&
                     L.nuc_dip_off + m.nsub <= np && L.time_off >= 0 && L.time_off + m.nsub <= np,
                 BEATAMD_EINVAL, "layout: hypocentre variables outside q");
    }
    BA_CHECK(!m.wavemaps.empty() || m.has_geo || m.lap >= 0, BEATAMD_EINVAL,
             "model has no composite");
    return BEATAMD_OK;
}

int beatamd_ffi_synthetics_batch(beatamd_ctx *ctx, int32_t model_id, int32_t wavemap_index, int64_t C,
                                 const double *Q, int32_t residuals, double *out)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    BA_CHECK(Q && out && C >= 0, BEATAMD_EINVAL, "ffi_synthetics: bad argument");
    BA_CHECK(wavemap_index >= 0 && (size_t)wavemap_index < m->wavemaps.size(), BEATAMD_EINVAL,
             "ffi_synthetics: model has no wavemap %d", wavemap_index);
    BA_TRY(model_check_layout(*m));
    if (C == 0) return BEATAMD_OK;
    Wavemap &wm = m->wavemaps[wavemap_index];
    const int64_t np = m->layout.nparams;
    const void *d_q;
    void *d_o, *p;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, Q, (size_t)C * np * 8, &d_q));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)C * wm.T * wm.N * 8, &d_o, &rec));
    const double *Qd = (const double *)d_q;
    BA_TRY(ctx->get_scratch(SL_CHAINBAD, (size_t)C * sizeof(int32_t), &p));
    int32_t *chain_bad = (int32_t *)p;
    BA_HIP(hipMemsetAsync(chain_bad, 0, (size_t)C * sizeof(int32_t), ctx->stream));
    BA_TRY(ctx->get_scratch(SL_ST0, (size_t)C * m->P * sizeof(double), &p));
    double *st0 = (double *)p;
    BA_TRY(launch_sweep_model(ctx, *m, Qd, C, st0, chain_bad));
    GfStackCall k;
    k.nvar = m->layout.nvar;
    for (int v = 0; v < k.nvar; v++) {
        k.libs[v] = get_obj(ctx->seislibs, wm.libs[v]);
        BA_CHECK(k.libs[v] && k.libs[v]->g, BEATAMD_EINVAL, "wavemap refers to a destroyed / empty GF library");
        k.slips[v] = ChainVec{Qd, np, m->layout.slip_off[v]};
    }
    k.durations = ChainVec{Qd, np, m->layout.durations_off};
    k.st.starttimes0 = st0;
    k.st.Q = Qd;
    k.st.nparams = np;
    k.order_key[0] = ChainVec{Qd, np, m->layout.nuc_strike_off};
    k.order_key[1] = ChainVec{Qd, np, m->layout.nuc_dip_off};
    k.st.shift_off = wm.shift_off;
    k.st.chain_bad = chain_bad;
    k.interp = wm.interp;
    k.C = C;
    k.data = wm.data;
    k.mode = residuals ? GF_RESID_STORE : GF_STORE_SYN;
    k.out = (double *)d_o;
    BA_TRY(launch_gfstack(ctx, k));
    BA_TRY(ctx->check_status());   // an index outside the library is the reference's IndexError
    return finish_out(ctx, &rec, 1);
}

int beatamd_ffi_logp_batch(beatamd_ctx *ctx, int32_t model_id, int64_t C, const double *Q,
                           double *LL)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    BA_CHECK(Q && LL && C >= 0, BEATAMD_EINVAL, "ffi_logp: bad argument");
    BA_TRY(model_check_layout(*m));
    if (C == 0) return BEATAMD_OK;
    const void *d_q;
    void *d_l;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, Q, (size_t)C * m->layout.nparams * 8, &d_q));
    BA_TRY(stage_out(ctx, SL_OUT0, LL, (size_t)C * m->nllk() * 8, &d_l, &rec));
    BA_TRY(ffi_logp_device(ctx, *m, C, (const double *)d_q, (double *)d_l));
    return finish_out(ctx, &rec, 1);
}

// proposal source of a step: rows handed in (delta, log_u) or drawn here (factor / scales + Philox key)
struct StepDraw {
    const double *factor = nullptr;   // [K, np] or the per-parameter scales [np]
    int64_t K = 0;
    int32_t kind = -1, df = 0;
    uint64_t seed = 0, first_chain = 0;
    uint32_t step = 0;
};

static int astep_impl(beatamd_ctx *ctx, int32_t model_id, int64_t C, double *Q0, double *L0,
                      const double *delta, const double *scaling, const double *lower,
                      const double *upper, const double *log_u, double beta, const double *betas,
                      int32_t *accepted, const StepDraw *draw = nullptr, int32_t *acc_sum = nullptr,
                      int64_t *n_acc = nullptr)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    BA_CHECK(Q0 && L0 && scaling && lower && upper && accepted && C >= 0 && (draw || (delta && log_u)),
             BEATAMD_EINVAL, "ffi_astep: NULL argument");
    BA_TRY(model_check_layout(*m));
    if (C == 0) return BEATAMD_OK;
    const int64_t np = m->layout.nparams, nllk = m->nllk();
    const void *d_de = nullptr, *d_sc, *d_lo, *d_up, *d_lu = nullptr, *d_be = nullptr, *d_f = nullptr;
    void *d_q0, *d_l0, *d_acc, *p;
    Arg recs[3];
    BA_TRY(stage_out(ctx, SL_OUT0, Q0, (size_t)C * np * 8, &d_q0, &recs[0], true));
    BA_TRY(stage_out(ctx, SL_OUT1, L0, (size_t)C * nllk * 8, &d_l0, &recs[1], true));
    BA_TRY(stage_out(ctx, SL_OUT2, accepted, (size_t)C * 4, &d_acc, &recs[2]));
    BA_TRY(stage_in(ctx, SL_IN2, scaling, (size_t)C * 8, &d_sc));
    BA_TRY(stage_in(ctx, SL_IN3, lower, (size_t)np * 8, &d_lo));
    BA_TRY(stage_in(ctx, SL_IN4, upper, (size_t)np * 8, &d_up));
    if (betas) BA_TRY(stage_in(ctx, SL_IN6, betas, (size_t)C * 8, &d_be));
    BA_TRY(ctx->get_scratch(SL_QPROP, (size_t)C * np * 8, &p));
    double *qprop = (double *)p;
    BA_TRY(ctx->get_scratch(SL_LPROP, (size_t)C * nllk * 8, &p));
    double *lprop = (double *)p;
    BA_TRY(ctx->get_scratch(SL_MISC, (size_t)C * 4 + 64, &p));
    int32_t *inb = (int32_t *)p;
    bool advance = false;
    if (draw) {
        BA_CHECK(is_device_ptr(Q0) && (!acc_sum || is_device_ptr(acc_sum)) && (!n_acc || is_device_ptr(n_acc)),
                 BEATAMD_EINVAL, "ffi_mstep: chain states and counters live on the device");
        const int64_t K = draw->kind < 0 ? draw->K : np;
        BA_TRY(stage_in(ctx, SL_IN0, draw->factor, (size_t)(draw->kind < 0 ? K * np : np) * 8, &d_f));
        BA_TRY(ctx->get_scratch(SL_LOGU, (size_t)C * 8, &p));
        double *lu = (double *)p;
        d_lu = lu;
        if (draw_propose_applicable(K, np)) {
            BA_TRY(launch_draw_propose(ctx, C, K, np, draw->kind, (const double *)d_f, draw->df, draw->seed,
                                       draw->step, draw->first_chain, (const double *)d_q0, (const double *)d_sc,
                                       (const double *)d_lo, (const double *)d_up, qprop, lu, inb));
        } else {
            BA_TRY(ctx->get_scratch(SL_DELTA, (size_t)C * np * 8, &p));
            double *de = (double *)p;
            if (draw->kind < 0) {
                BA_TRY(ctx->get_scratch(SL_Z, (size_t)C * K * 8, &p));
                double *z = (double *)p, *rs = nullptr;
                if (draw->df > 0) {
                    BA_TRY(ctx->get_scratch(SL_ROWSCALE, (size_t)C * 8, &p));
                    rs = (double *)p;
                }
                BA_TRY(launch_philox_normal(ctx, z, C, K, draw->seed, draw->step, draw->first_chain));
                BA_TRY(launch_philox_chain(ctx, C, draw->seed, draw->step, draw->first_chain, draw->df, lu, rs));
                GemmCall g;
                g.A = z; g.lda = K;
                g.B = (const double *)d_f; g.ldb = np; g.b_kn = 1;
                g.O = de; g.ldo = np;
                g.M = C; g.N = np; g.K = K;
                g.row_scale = rs;
                g.timer = "proposal";
                BA_TRY(launch_gemm_f64(ctx, g));
            } else {
                BA_TRY(launch_philox_univariate(ctx, de, C, np, draw->kind, (const double *)d_f, draw->seed,
                                                draw->step, draw->first_chain));
                BA_TRY(launch_philox_chain(ctx, C, draw->seed, draw->step, draw->first_chain, 0, lu, nullptr));
            }
            BA_TRY(launch_propose(ctx, C, np, (const double *)d_q0, de, (const double *)d_sc,
                                  (const double *)d_lo, (const double *)d_up, qprop, inb));
        }
        advance = true;
    } else {
        BA_TRY(stage_in(ctx, SL_IN1, delta, (size_t)C * np * 8, &d_de));
        BA_TRY(stage_in(ctx, SL_IN5, log_u, (size_t)C * 8, &d_lu));
        BA_TRY(launch_propose(ctx, C, np, (const double *)d_q0, (const double *)d_de,
                              (const double *)d_sc, (const double *)d_lo, (const double *)d_up, qprop,
                              inb));
    }
    // the `like` sum rides in the accept kernel (one launch fewer) while the row fits its LDS stage
    LikeTail tail;
    const bool fold = nllk * 8 <= 48 * 1024;
    BA_TRY(ffi_logp_device(ctx, *m, C, qprop, lprop, fold ? &tail : nullptr));
    BA_TRY(launch_accept(ctx, C, np, nllk, (double *)d_q0, (double *)d_l0, qprop, lprop, inb,
                         (const double *)d_lu, beta, (const double *)d_be, (int32_t *)d_acc,
                         fold ? &tail.grp : nullptr, tail.chain_bad, acc_sum, n_acc, advance));
    return finish_out(ctx, recs, 3);
}

int beatamd_ffi_astep_batch(beatamd_ctx *ctx, int32_t model_id, int64_t C, double *Q0, double *L0,
                            const double *delta, const double *scaling, const double *lower,
                            const double *upper, const double *log_u, double beta,
                            int32_t *accepted)
{
    BA_CHECK(ctx != nullptr, BEATAMD_EINVAL, "ctx is NULL");
    return astep_impl(ctx, model_id, C, Q0, L0, delta, scaling, lower, upper, log_u, beta, nullptr,
                      accepted);
}

int beatamd_ffi_astep_batch_betas(beatamd_ctx *ctx, int32_t model_id, int64_t C, double *Q0,
                                  double *L0, const double *delta, const double *scaling,
                                  const double *lower, const double *upper, const double *log_u,
                                  const double *betas, int32_t *accepted)
{
    BA_CHECK(ctx != nullptr && betas != nullptr, BEATAMD_EINVAL, "ffi_astep_betas: NULL argument");
    return astep_impl(ctx, model_id, C, Q0, L0, delta, scaling, lower, upper, log_u, 1.0, betas,
                      accepted);
}

int beatamd_ffi_mstep_batch(beatamd_ctx *ctx, int32_t model_id, int64_t C, double *Q0, double *L0,
                            const double *factor, int64_t K, int32_t kind, int32_t df, uint64_t seed,
                            uint32_t step, int64_t first_chain, const double *scaling, const double *lower,
                            const double *upper, double beta, const double *betas, int32_t *accepted,
                            int32_t *accepted_sum, int64_t *n_accepted)
{
    BA_CHECK(ctx != nullptr, BEATAMD_EINVAL, "ctx is NULL");
    BA_CHECK(factor && first_chain >= 0 && df >= 0 && df <= 64, BEATAMD_EINVAL, "ffi_mstep: bad argument");
    BA_CHECK(kind >= -1 && kind <= BEATAMD_PROPOSAL_POISSON && (kind >= 0 || K > 0), BEATAMD_EINVAL,
             "ffi_mstep: kind must be -1 (multivariate, K > 0 factor rows), Normal (0), Cauchy (1), Laplace (2) or "
             "Poisson (3)");
    StepDraw d;
    d.factor = factor; d.K = K; d.kind = kind; d.df = kind < 0 ? df : 0;
    d.seed = seed; d.step = step; d.first_chain = (uint64_t)first_chain;
    return astep_impl(ctx, model_id, C, Q0, L0, nullptr, scaling, lower, upper, nullptr, betas ? 1.0 : beta,
                      betas, accepted, &d, accepted_sum, n_accepted);
}

// ------------------------------------------------------------------ noise covariance
int beatamd_autocovariance_batch(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *data,
                                 const double *mean, double *out)
{
    ENTER(ctx);
    BA_CHECK(data && mean && out && nd >= 0 && n >= 0, BEATAMD_EINVAL, "autocovariance: bad argument");
    if (nd == 0 || n == 0) return BEATAMD_OK;
    const void *d_d, *d_m;
    void *d_o;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, data, (size_t)nd * n * 8, &d_d));
    BA_TRY(stage_in(ctx, SL_IN1, mean, (size_t)nd * 8, &d_m));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)nd * n * 8, &d_o, &rec));
    BA_TRY(launch_autocovariance(ctx, nd, n, (const double *)d_d, (const double *)d_m, (double *)d_o));
    return finish_out(ctx, &rec, 1);
}

int beatamd_scaled_toeplitz_batch(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *coeffs,
                                  const double *stds, double *out)
{
    ENTER(ctx);
    BA_CHECK(coeffs && stds && out && nd >= 0 && n >= 0, BEATAMD_EINVAL, "scaled_toeplitz: bad argument");
    if (nd == 0 || n == 0) return BEATAMD_OK;
    const void *d_c, *d_s;
    void *d_o;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, coeffs, (size_t)nd * n * 8, &d_c));
    BA_TRY(stage_in(ctx, SL_IN1, stds, (size_t)nd * n * 8, &d_s));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)nd * n * n * 8, &d_o, &rec));
    BA_TRY(launch_scaled_toeplitz(ctx, nd, n, (const double *)d_c, (const double *)d_s, (double *)d_o));
    return finish_out(ctx, &rec, 1);
}

// ------------------------------------------------------------------ introspection
int beatamd_ctx_last_kernel(beatamd_ctx *ctx, char *buf, int64_t buflen)
{
    BA_CHECK(ctx && buf && buflen > 0, BEATAMD_EINVAL, "last_kernel: bad argument");
    snprintf(buf, (size_t)buflen, "%s", ctx->last_gf_kernel);
    return BEATAMD_OK;
}

int beatamd_ctx_gf_group_stats(beatamd_ctx *ctx, int64_t *chains_per_group, double *mean_rows,
                               int64_t *max_rows, int64_t *row_bytes)
{
    ENTER(ctx);
    BA_CHECK(chains_per_group && mean_rows && max_rows && row_bytes, BEATAMD_EINVAL,
             "gf_group_stats: NULL argument");
    *chains_per_group = ctx->gs_cg;
    *mean_rows = 0.0;
    *max_rows = 0;
    *row_bytes = 0;
    if (ctx->gs_ngtp == 0) return BEATAMD_OK;  // the last launch was the streaming kernel
    BA_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> uc((size_t)ctx->gs_ngtp);
    BA_HIP(hipMemcpy(uc.data(), ctx->scratch[SL_GS_UCOUNT].p, uc.size() * sizeof(uint32_t),
                     hipMemcpyDeviceToHost));
    int64_t tot = 0, mx = 0;
    for (uint32_t u : uc) {
        tot += u;
        mx = std::max<int64_t>(mx, u);
    }
    *mean_rows = (double)tot / (double)uc.size();
    *max_rows = mx;
    // every distinct row is staged once per (group, target) and slip variable
    *row_bytes = (int64_t)((double)tot * ctx->gs_trep) * ctx->gs_N * 8 * ctx->gs_nvar;
    return BEATAMD_OK;
}

int beatamd_ctx_gf_plan(beatamd_ctx *ctx, char *buf, int64_t buflen, double *mean_passes, int64_t *max_passes)
{
    ENTER(ctx);
    BA_CHECK(buf && buflen > 0, BEATAMD_EINVAL, "gf_plan: bad argument");
    snprintf(buf, (size_t)buflen, "%s", ctx->gf_plan);
    if (mean_passes) *mean_passes = ctx->gs_ngtp ? 1.0 : 0.0;
    if (max_passes) *max_passes = ctx->gs_ngtp ? 1 : 0;
    if (!ctx->gs_has_passes || ctx->gs_ngtp == 0 || (!mean_passes && !max_passes)) return BEATAMD_OK;
    BA_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> np((size_t)ctx->gs_ngtp);
    BA_HIP(hipMemcpy(np.data(), (const uint32_t *)ctx->scratch[SL_GS_UCOUNT].p + ctx->gs_ngtp, np.size() * sizeof(uint32_t),
                     hipMemcpyDeviceToHost));
    int64_t tot = 0, mx = 0;
    for (uint32_t u : np) {
        tot += u;
        mx = std::max<int64_t>(mx, u);
    }
    if (mean_passes) *mean_passes = (double)tot / (double)np.size();
    if (max_passes) *max_passes = mx;
    return BEATAMD_OK;
}

int beatamd_ctx_gf_tune_log(beatamd_ctx *ctx, char *buf, int64_t buflen)
{
    ENTER(ctx);
    BA_CHECK(buf && buflen > 0, BEATAMD_EINVAL, "gf_tune_log: bad argument");
    snprintf(buf, (size_t)buflen, "%s", ctx->gf_tune_log);
    return BEATAMD_OK;
}

int32_t beatamd_gf_patch_ranges(int64_t ntargets, int64_t npatches, int64_t nsamples, int32_t num_cu)
{
    if (ntargets <= 0 || npatches <= 0 || nsamples <= 0) return 1;
    return (int32_t)gf_patch_ranges(ntargets, npatches, nsamples, num_cu > 0 ? num_cu : 256);
}

int beatamd_ctx_gf_chain_groups(beatamd_ctx *ctx, int64_t C, const double *key0, const double *key1,
                                int64_t chains_per_group, uint32_t *members)
{
    ENTER(ctx);
    BA_CHECK(C > 0 && key0 && key1 && members && chains_per_group > 0, BEATAMD_EINVAL, "gf_chain_groups: bad argument");
    const int64_t ngroups = (C + chains_per_group - 1) / chains_per_group;
    const ChainVec key[2] = {ChainVec{key0, 1, 0}, ChainVec{key1, 1, 0}};
    const uint32_t *m = nullptr;
    BA_TRY(launch_chain_members(ctx, C, key, chains_per_group, ngroups, &m));
    BA_CHECK(m != nullptr, BEATAMD_EINVAL, "gf_chain_groups: groups of more than 8192 chains are not cut");
    BA_HIP(hipStreamSynchronize(ctx->stream));
    BA_HIP(hipMemcpy(members, m, (size_t)(ngroups * chains_per_group) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BEATAMD_OK;
}

// ------------------------------------------------------------------ SMC stage transition
int beatamd_smc_calc_beta(beatamd_ctx *ctx, int64_t C, const double *likelihoods, int64_t stride,
                          double beta, double coef_variation, double *beta_new, double *weights)
{
    ENTER(ctx);
    BA_CHECK(likelihoods && beta_new && weights && C > 0 && stride > 0, BEATAMD_EINVAL,
             "smc_calc_beta: bad argument");
    const void *d_l;
    void *d_w, *p;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, likelihoods, (size_t)((C - 1) * stride + 1) * 8, &d_l));
    BA_TRY(stage_out(ctx, SL_OUT0, weights, (size_t)C * 8, &d_w, &rec));
    BA_TRY(ctx->get_scratch(SL_STAGE2, 64, &p));
    BA_TRY(launch_smc_calc_beta(ctx, C, (const double *)d_l, stride, beta, coef_variation, 0, 0.0,
                                (double *)p, (double *)d_w));
    double out[2];
    BA_HIP(hipMemcpyAsync(out, p, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    BA_TRY(finish_out(ctx, &rec, 1));
    BA_HIP(hipStreamSynchronize(ctx->stream));
    *beta_new = out[0];
    return BEATAMD_OK;
}

int beatamd_smc_stage_weights(beatamd_ctx *ctx, int64_t C, const double *likelihoods, int64_t stride,
                              double dbeta, double *weights)
{
    ENTER(ctx);
    BA_CHECK(likelihoods && weights && C > 0 && stride > 0, BEATAMD_EINVAL, "smc_stage_weights: bad argument");
    const void *d_l;
    void *d_w, *p;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, likelihoods, (size_t)((C - 1) * stride + 1) * 8, &d_l));
    BA_TRY(stage_out(ctx, SL_OUT0, weights, (size_t)C * 8, &d_w, &rec));
    BA_TRY(ctx->get_scratch(SL_STAGE2, 64, &p));
    BA_TRY(launch_smc_calc_beta(ctx, C, (const double *)d_l, stride, 0.0, 0.0, 1, dbeta, (double *)p,
                                (double *)d_w));
    return finish_out(ctx, &rec, 1);
}

int beatamd_smc_resample(beatamd_ctx *ctx, int64_t C, const double *weights, double aux,
                         int32_t *indexes)
{
    ENTER(ctx);
    BA_CHECK(weights && indexes && C > 0, BEATAMD_EINVAL, "smc_resample: bad argument");
    BA_CHECK(C < (int64_t)0x7fffffff, BEATAMD_EINVAL, "smc_resample: too many chains");
    const void *d_w;
    void *d_i, *p;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, weights, (size_t)C * 8, &d_w));
    BA_TRY(stage_out(ctx, SL_OUT0, indexes, (size_t)C * 4, &d_i, &rec));
    BA_TRY(ctx->get_scratch(SL_CUM, (size_t)C * 8, &p));
    BA_TRY(launch_smc_resample(ctx, C, (const double *)d_w, aux, (double *)p, (int32_t *)d_i));
    return finish_out(ctx, &rec, 1);
}

int beatamd_smc_population_factor(beatamd_ctx *ctx, int64_t C, int64_t nparams,
                                  const double *population, const double *weights, double *factor)
{
    ENTER(ctx);
    BA_CHECK(population && weights && factor && C > 0 && nparams > 0, BEATAMD_EINVAL,
             "smc_population_factor: bad argument");
    const void *d_x, *d_w;
    void *d_f;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, population, (size_t)C * nparams * 8, &d_x));
    BA_TRY(stage_in(ctx, SL_IN1, weights, (size_t)C * 8, &d_w));
    BA_TRY(stage_out(ctx, SL_OUT0, factor, (size_t)C * nparams * 8, &d_f, &rec));
    BA_TRY(launch_pop_factor(ctx, C, nparams, (const double *)d_x, nparams, (const double *)d_w,
                             (double *)d_f));
    return finish_out(ctx, &rec, 1);
}

int beatamd_proposal_draw(beatamd_ctx *ctx, int64_t C, int64_t K, int64_t nparams,
                          const double *factor, uint64_t seed, uint32_t step, int64_t first_chain,
                          int32_t df, double *delta, double *log_u)
{
    ENTER(ctx);
    BA_CHECK(factor && delta && C >= 0 && K > 0 && nparams > 0 && df >= 0 && first_chain >= 0,
             BEATAMD_EINVAL, "proposal_draw: bad argument");
    BA_CHECK(df <= 64, BEATAMD_EINVAL, "proposal_draw: at most 64 degrees of freedom");
    if (C == 0) return BEATAMD_OK;
    const void *d_f;
    void *d_d, *d_u = nullptr, *p;
    Arg recs[2];
    int nrec = 1;
    BA_TRY(stage_in(ctx, SL_IN0, factor, (size_t)K * nparams * 8, &d_f));
    BA_TRY(stage_out(ctx, SL_OUT0, delta, (size_t)C * nparams * 8, &d_d, &recs[0]));
    if (log_u) {
        BA_TRY(stage_out(ctx, SL_OUT1, log_u, (size_t)C * 8, &d_u, &recs[1]));
        nrec = 2;
    }
    BA_TRY(ctx->get_scratch(SL_Z, (size_t)C * K * 8, &p));
    double *z = (double *)p;
    double *rs = nullptr;
    if (df > 0) {
        BA_TRY(ctx->get_scratch(SL_ROWSCALE, (size_t)C * 8, &p));
        rs = (double *)p;
    }
    BA_TRY(launch_philox_normal(ctx, z, C, K, seed, step, (uint64_t)first_chain));
    if (d_u || rs)
        BA_TRY(launch_philox_chain(ctx, C, seed, step, (uint64_t)first_chain, df, (double *)d_u, rs));
    GemmCall g;
    g.A = z; g.lda = K;
    g.B = (const double *)d_f; g.ldb = nparams; g.b_kn = 1;
    g.O = (double *)d_d; g.ldo = nparams;
    g.M = C; g.N = nparams; g.K = K;
    g.row_scale = rs;
    g.timer = "proposal";
    BA_TRY(launch_gemm_f64(ctx, g));
    BA_TRY(launch_step_advance(ctx));
    return finish_out(ctx, recs, nrec);
}

int beatamd_proposal_draw_univariate(beatamd_ctx *ctx, int64_t C, int64_t nparams, int32_t kind,
                                     const double *scale, uint64_t seed, uint32_t step, int64_t first_chain,
                                     double *delta, double *log_u)
{
    ENTER(ctx);
    BA_CHECK(scale && delta && C >= 0 && nparams > 0 && first_chain >= 0, BEATAMD_EINVAL,
             "proposal_draw_univariate: bad argument");
    BA_CHECK(kind >= BEATAMD_PROPOSAL_NORMAL && kind <= BEATAMD_PROPOSAL_POISSON, BEATAMD_EINVAL,
             "proposal_draw_univariate: kind must be Normal (0), Cauchy (1), Laplace (2) or Poisson (3)");
    if (C == 0) return BEATAMD_OK;
    const void *d_s;
    void *d_d, *d_u = nullptr;
    Arg recs[2];
    int nrec = 1;
    BA_TRY(stage_in(ctx, SL_IN0, scale, (size_t)nparams * 8, &d_s));
    BA_TRY(stage_out(ctx, SL_OUT0, delta, (size_t)C * nparams * 8, &d_d, &recs[0]));
    if (log_u) {
        BA_TRY(stage_out(ctx, SL_OUT1, log_u, (size_t)C * 8, &d_u, &recs[1]));
        nrec = 2;
    }
    BA_TRY(launch_philox_univariate(ctx, (double *)d_d, C, nparams, kind, (const double *)d_s, seed, step,
                                    (uint64_t)first_chain));
    if (d_u) BA_TRY(launch_philox_chain(ctx, C, seed, step, (uint64_t)first_chain, 0, (double *)d_u, nullptr));
    BA_TRY(launch_step_advance(ctx));
    return finish_out(ctx, recs, nrec);
}

int beatamd_gather_rows(beatamd_ctx *ctx, int64_t nout, int64_t ncols, const double *src,
                        int64_t nrows_src, const int32_t *indexes, double *out)
{
    ENTER(ctx);
    BA_CHECK(src && indexes && out && nout >= 0 && ncols > 0 && nrows_src > 0, BEATAMD_EINVAL,
             "gather_rows: bad argument");
    if (nout == 0) return BEATAMD_OK;
    const void *d_s, *d_i;
    void *d_o;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, src, (size_t)nrows_src * ncols * 8, &d_s));
    BA_TRY(stage_in(ctx, SL_IN1, indexes, (size_t)nout * 4, &d_i));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)nout * ncols * 8, &d_o, &rec));
    BA_CHECK(d_o != d_s, BEATAMD_EINVAL, "gather_rows: out must not alias src");
    BA_TRY(launch_gather_rows(ctx, nout, ncols, (const double *)d_s, ncols, nrows_src,
                              (const int32_t *)d_i, (double *)d_o, ncols));
    return finish_out(ctx, &rec, 1);
}

// ------------------------------------------------------------------ a Metropolis step in pieces (target-sharded models)
int beatamd_like_assemble(beatamd_ctx *ctx, int64_t C, int64_t nllk, int64_t nsrc, const double *gathered,
                          const int32_t *dst_col, const double *local_ll, int64_t local_ld, int64_t local_col0,
                          int64_t n_rest, int64_t rest_dst0, int32_t ngroups, const int32_t *group_end, double *LL)
{
    ENTER(ctx);
    BA_CHECK(gathered && dst_col && LL && group_end && C >= 0 && nsrc > 0 && nllk > 1 && ngroups >= 1 && ngroups <= 8 &&
             n_rest >= 0 && (n_rest == 0 || local_ll), BEATAMD_EINVAL, "like_assemble: bad argument");
    BA_CHECK(is_device_ptr(gathered) && is_device_ptr(LL) && (!local_ll || is_device_ptr(local_ll)), BEATAMD_EINVAL,
             "like_assemble: the gathered block, the local vectors and LL live on the device");
    if (C == 0) return BEATAMD_OK;
    LikeGroups grp;
    grp.n = ngroups;
    for (int g = 0; g < ngroups; g++) {
        BA_CHECK(group_end[g] > 0 && group_end[g] < nllk && (g == 0 || group_end[g] >= group_end[g - 1]), BEATAMD_EINVAL,
                 "like_assemble: composite boundaries must ascend inside the vector");
        grp.end[g] = group_end[g];
    }
    for (int64_t r = 0; r < nsrc; r++)
        BA_CHECK(dst_col[r] >= -1 && dst_col[r] < nllk - 1, BEATAMD_EINVAL, "like_assemble: column %d of row %lld", dst_col[r], (long long)r);
    BA_CHECK(rest_dst0 >= 0 && rest_dst0 + n_rest <= nllk - 1, BEATAMD_EINVAL, "like_assemble: replicated columns outside the vector");
    const void *d_dc;
    BA_TRY(stage_in(ctx, SL_IN0, dst_col, (size_t)nsrc * 4, &d_dc));
    BA_TRY(launch_like_assemble(ctx, C, nllk, nsrc, gathered, (const int32_t *)d_dc, local_ll, local_ld, local_col0, n_rest,
                                rest_dst0, grp, LL, nullptr));
    return BEATAMD_OK;
}

int beatamd_metropolis_propose(beatamd_ctx *ctx, int64_t C, int64_t nparams, const double *Q0, const double *delta,
                               const double *scaling, const double *lower, const double *upper, double *Qprop,
                               int32_t *inbounds)
{
    ENTER(ctx);
    BA_CHECK(Q0 && delta && scaling && lower && upper && Qprop && inbounds && C >= 0 && nparams > 0, BEATAMD_EINVAL,
             "metropolis_propose: bad argument");
    BA_CHECK(is_device_ptr(Q0) && is_device_ptr(delta) && is_device_ptr(scaling) && is_device_ptr(lower) && is_device_ptr(upper) &&
             is_device_ptr(Qprop) && is_device_ptr(inbounds), BEATAMD_EINVAL, "metropolis_propose: device pointers only");
    return launch_propose(ctx, C, nparams, Q0, delta, scaling, lower, upper, Qprop, inbounds);
}

int beatamd_metropolis_accept(beatamd_ctx *ctx, int64_t C, int64_t nparams, int64_t nllk, double *Q0, double *L0,
                              const double *Qprop, double *Lprop, const int32_t *inbounds, const double *log_u, double beta,
                              const double *betas, int32_t *accepted)
{
    ENTER(ctx);
    BA_CHECK(Q0 && L0 && Qprop && Lprop && inbounds && log_u && accepted && C >= 0 && nparams > 0 && nllk > 0, BEATAMD_EINVAL,
             "metropolis_accept: bad argument");
    BA_CHECK(is_device_ptr(Q0) && is_device_ptr(L0) && is_device_ptr(Qprop) && is_device_ptr(Lprop) && is_device_ptr(inbounds) &&
             is_device_ptr(log_u) && is_device_ptr(accepted) && (!betas || is_device_ptr(betas)), BEATAMD_EINVAL,
             "metropolis_accept: device pointers only");
    return launch_accept(ctx, C, nparams, nllk, Q0, L0, Qprop, Lprop, inbounds, log_u, beta, betas, accepted, nullptr, nullptr,
                         nullptr, nullptr, false);
}

int beatamd_metropolis_tune(beatamd_ctx *ctx, int64_t C, double *scaling, int32_t *accepted,
                            int32_t tune_interval)
{
    ENTER(ctx);
    BA_CHECK(scaling && accepted && C >= 0 && tune_interval > 0, BEATAMD_EINVAL,
             "metropolis_tune: bad argument");
    if (C == 0) return BEATAMD_OK;
    void *d_s, *d_a;
    Arg recs[2];
    BA_TRY(stage_out(ctx, SL_OUT0, scaling, (size_t)C * 8, &d_s, &recs[0], true));
    BA_TRY(stage_out(ctx, SL_OUT1, accepted, (size_t)C * 4, &d_a, &recs[1], true));
    BA_TRY(launch_tune_scaling(ctx, C, (double *)d_s, (int32_t *)d_a, (double)tune_interval));
    return finish_out(ctx, recs, 2);
}

// ------------------------------------------------------------------ whitening operator
int beatamd_chol_inverse_batch(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *covs, double *W,
                               double *log_pdet)
{
    ENTER(ctx);
    BA_CHECK(covs && W && log_pdet && nd >= 0 && n > 0, BEATAMD_EINVAL, "chol_inverse_batch: bad argument");
    if (nd == 0) return BEATAMD_OK;
    const void *d_c;
    void *d_w, *d_l;
    Arg recs[2];
    BA_TRY(stage_in(ctx, SL_IN0, covs, (size_t)nd * n * n * 8, &d_c));
    BA_TRY(stage_out(ctx, SL_OUT0, W, (size_t)nd * n * n * 8, &d_w, &recs[0]));
    BA_TRY(stage_out(ctx, SL_OUT1, log_pdet, (size_t)nd * 8, &d_l, &recs[1]));
    BA_TRY(launch_chol_inverse(ctx, nd, n, (const double *)d_c, (double *)d_w, (double *)d_l));
    BA_TRY(ctx->check_status());
    return finish_out(ctx, recs, 2);
}

int beatamd_chol_inverse_batch_flags(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *covs, double *W,
                                     double *log_pdet, int32_t *not_psd)
{
    ENTER(ctx);
    BA_CHECK(covs && W && log_pdet && not_psd && nd >= 0 && n > 0, BEATAMD_EINVAL,
             "chol_inverse_batch_flags: bad argument");
    if (nd == 0) return BEATAMD_OK;
    const void *d_c;
    void *d_w, *d_l, *d_f;
    Arg recs[3];
    BA_TRY(stage_in(ctx, SL_IN0, covs, (size_t)nd * n * n * 8, &d_c));
    BA_TRY(stage_out(ctx, SL_OUT0, W, (size_t)nd * n * n * 8, &d_w, &recs[0]));
    BA_TRY(stage_out(ctx, SL_OUT1, log_pdet, (size_t)nd * 8, &d_l, &recs[1]));
    BA_TRY(stage_out(ctx, SL_OUT2, not_psd, (size_t)nd * 4, &d_f, &recs[2]));
    BA_TRY(launch_chol_inverse(ctx, nd, n, (const double *)d_c, (double *)d_w, (double *)d_l, (int32_t *)d_f));
    return finish_out(ctx, recs, 3);
}

int beatamd_factor_compact(beatamd_ctx *ctx, int64_t K, int64_t n, const double *factor, double *R)
{
    ENTER(ctx);
    BA_CHECK(factor && R && K > 0 && n > 0, BEATAMD_EINVAL, "factor_compact: bad argument");
    const void *d_f;
    void *d_r;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, factor, (size_t)K * n * 8, &d_f));
    BA_TRY(stage_out(ctx, SL_OUT0, R, (size_t)n * n * 8, &d_r, &rec));
    BA_TRY(launch_gram_cholesky(ctx, K, n, (const double *)d_f, (double *)d_r));
    BA_TRY(ctx->check_status());
    return finish_out(ctx, &rec, 1);
}

int beatamd_whitening_ratio_batch(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *W_new,
                                  const double *W_old, double *M)
{
    ENTER(ctx);
    BA_CHECK(W_new && W_old && M && nd >= 0 && n > 0, BEATAMD_EINVAL, "whitening_ratio_batch: bad argument");
    if (nd == 0) return BEATAMD_OK;
    const void *d_n, *d_o;
    void *d_m;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, W_new, (size_t)nd * n * n * 8, &d_n));
    BA_TRY(stage_in(ctx, SL_IN1, W_old, (size_t)nd * n * n * 8, &d_o));
    BA_TRY(stage_out(ctx, SL_OUT0, M, (size_t)nd * n * n * 8, &d_m, &rec));
    BA_TRY(launch_triu_ratio(ctx, nd, n, (const double *)d_n, (const double *)d_o, (double *)d_m));
    BA_TRY(ctx->check_status());
    return finish_out(ctx, &rec, 1);
}

int beatamd_unwhiten_traces(beatamd_ctx *ctx, int64_t nd, int64_t n, const double *W, double *X)
{
    ENTER(ctx);
    BA_CHECK(W && X && nd >= 0 && n > 0, BEATAMD_EINVAL, "unwhiten_traces: bad argument");
    if (nd == 0) return BEATAMD_OK;
    const void *d_w;
    void *d_o;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, W, (size_t)nd * n * n * 8, &d_w));
    BA_TRY(stage_out(ctx, SL_OUT0, X, (size_t)nd * n * 8, &d_o, &rec, true));   // in place: host traces go up first
    BA_TRY(launch_triu_solve_vec(ctx, nd, n, (const double *)d_w, (double *)d_o));
    BA_TRY(ctx->check_status());
    return finish_out(ctx, &rec, 1);
}

int beatamd_ffi_model_update_data(beatamd_ctx *ctx, int32_t model_id, int32_t wavemap_index, const double *data)
{
    ENTER(ctx);
    FfiModel *m = get_obj(ctx->models, model_id);
    BA_CHECK(m, BEATAMD_EINVAL, "unknown model %d", model_id);
    BA_CHECK(data && wavemap_index >= 0 && (size_t)wavemap_index < m->wavemaps.size(), BEATAMD_EINVAL,
             "model_update_data: bad argument");
    Wavemap &w = m->wavemaps[wavemap_index];
    BA_HIP(hipMemcpyAsync(w.data, data, (size_t)w.T * w.N * 8, hipMemcpyDefault, ctx->stream));
    BA_HIP(hipStreamSynchronize(ctx->stream));
    return BEATAMD_OK;
}

// ------------------------------------------------------------------ library whitening
int beatamd_whiten_rows(beatamd_ctx *ctx, double *rows, int64_t nrows, int64_t N, const double *W)
{
    ENTER(ctx);
    BA_CHECK(rows && W && nrows >= 0 && N > 0, BEATAMD_EINVAL, "whiten_rows: bad argument");
    BA_CHECK(is_device_ptr(rows), BEATAMD_EINVAL, "whiten_rows: rows must live in HBM");
    if (nrows == 0) return BEATAMD_OK;
    drop_f32_overlapping(ctx, rows, (size_t)nrows * N * 8);
    const void *d_w;
    void *p;
    BA_TRY(stage_in(ctx, SL_IN0, W, (size_t)N * N * 8, &d_w));
    BA_TRY(ctx->get_scratch(SL_MISC, 64, &p));
    BA_TRY(launch_check_upper_tri(ctx, (const double *)d_w, 1, N, (int *)p));
    int upper = 0;
    BA_HIP(hipMemcpyAsync(&upper, p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    BA_HIP(hipStreamSynchronize(ctx->stream));
    // out-of-place per chunk (every column block of a chunk reads all of its rows), then copied back
    int64_t chunk = std::max<int64_t>(64, ((int64_t)1 << 28) / (N * 8));
    chunk = std::min(chunk, nrows);
    BA_TRY(ctx->get_scratch(SL_WHITEN, (size_t)chunk * N * 8, &p));
    double *tmp = (double *)p;
    for (int64_t r0 = 0; r0 < nrows; r0 += chunk) {
        const int64_t nr = std::min(chunk, nrows - r0);
        GemmCall g;
        g.A = rows + r0 * N; g.lda = N;
        g.B = (const double *)d_w; g.ldb = N; g.b_kn = 0; g.b_upper = upper;
        g.O = tmp; g.ldo = N;
        g.M = nr; g.N = N; g.K = N;
        g.timer = "whiten";
        BA_TRY(launch_gemm_f64(ctx, g));
        BA_HIP(hipMemcpyAsync(rows + r0 * N, tmp, (size_t)nr * N * 8, hipMemcpyDeviceToDevice,
                              ctx->stream));
    }
    return BEATAMD_OK;
}

// All datasets of a wavemap at once, IN PLACE and without a second buffer: with upper-triangular operators
// (chol_inverse, heart.py:233) the product column n of a row needs the row's entries k >= n only, so the columns
// are produced 128 at a time in ascending order, each launch covering that column block of every row of every
// dataset (grid.y = dataset): what a block overwrites is never read by a later launch.  No chunk buffer, no
// copy back, no triangular load imbalance inside a launch, one host synchronisation (the triangularity check).
int beatamd_whiten_rows_batch(beatamd_ctx *ctx, double *rows, int64_t nbatch, int64_t nrows, int64_t N,
                              const double *W)
{
    ENTER(ctx);
    BA_CHECK(rows && W && nbatch >= 0 && nrows >= 0 && N > 0, BEATAMD_EINVAL, "whiten_rows_batch: bad argument");
    BA_CHECK(is_device_ptr(rows), BEATAMD_EINVAL, "whiten_rows_batch: rows must live in HBM");
    BA_CHECK(nbatch <= 65535, BEATAMD_EINVAL, "whiten_rows_batch: at most 65535 datasets per call");
    if (nrows == 0 || nbatch == 0) return BEATAMD_OK;
    drop_f32_overlapping(ctx, rows, (size_t)nbatch * nrows * N * 8);
    const void *d_w;
    void *p;
    BA_TRY(stage_in(ctx, SL_IN1, W, (size_t)nbatch * N * N * 8, &d_w));
    BA_TRY(ctx->get_scratch(SL_MISC, 64, &p));
    BA_TRY(launch_check_upper_tri(ctx, (const double *)d_w, nbatch, N, (int *)p));
    int upper = 0;
    BA_HIP(hipMemcpyAsync(&upper, p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    BA_HIP(hipStreamSynchronize(ctx->stream));
    if (!upper) {
        // general operators (a QR-fallback chol_inverse, heart.py:234-237): dataset by dataset through a buffer
        for (int64_t b = 0; b < nbatch; b++)
            BA_TRY(beatamd_whiten_rows(ctx, rows + b * nrows * N, nrows, N, (const double *)d_w + b * N * N));
        return BEATAMD_OK;
    }
    const int ncb = (int)((N + 127) / 128);
    for (int cb = 0; cb < ncb; cb++) {
        GemmCall g;
        g.A = rows; g.lda = N; g.sA = nrows * N;
        g.B = (const double *)d_w; g.ldb = N; g.sB = N * N; g.b_kn = 0; g.b_upper = 1;
        g.O = rows; g.ldo = N; g.sO = nrows * N;
        g.M = nrows; g.N = N; g.K = N;
        g.nbatch = (int)nbatch;
        g.col_block = cb;
        g.timer = "whiten";
        BA_TRY(launch_gemm_f64(ctx, g));
    }
    return BEATAMD_OK;
}

// ------------------------------------------------------------------ half-space synthetics
int beatamd_halfspace_displacements_batch(beatamd_ctx *ctx, int64_t C, int32_t nsrc,
                                          const int32_t *kind, const double *params, int64_t nobs,
                                          const double *east, const double *north, double nu,
                                          double *out)
{
    ENTER(ctx);
    BA_CHECK(kind && params && east && north && out && C >= 0 && nsrc > 0 && nobs > 0, BEATAMD_EINVAL,
             "halfspace_displacements: bad argument");
    BA_CHECK(nu > -1.0 && nu < 0.5, BEATAMD_EINVAL, "Poisson ratio outside (-1, 0.5)");
    BA_CHECK(!is_device_ptr(kind), BEATAMD_EINVAL, "halfspace_displacements: kind must be a host array");
    for (int i = 0; i < nsrc; i++)
        BA_CHECK(kind[i] == 0 || kind[i] == 1, BEATAMD_EINVAL, "unknown source kind %d", kind[i]);
    if (C == 0) return BEATAMD_OK;
    std::vector<int64_t> poff((size_t)nsrc * 10);
    for (size_t i = 0; i < poff.size(); i++) poff[i] = (int64_t)i;
    const void *d_k, *d_o, *d_p, *d_e, *d_n;
    void *d_out;
    Arg rec;
    BA_TRY(stage_in(ctx, SL_IN0, kind, (size_t)nsrc * 4, &d_k));
    BA_TRY(stage_in(ctx, SL_IN1, poff.data(), poff.size() * 8, &d_o));
    BA_TRY(stage_in(ctx, SL_IN2, params, (size_t)C * nsrc * 10 * 8, &d_p));
    BA_TRY(stage_in(ctx, SL_IN3, east, (size_t)nobs * 8, &d_e));
    BA_TRY(stage_in(ctx, SL_IN4, north, (size_t)nobs * 8, &d_n));
    BA_TRY(stage_out(ctx, SL_OUT0, out, (size_t)C * nsrc * nobs * 3 * 8, &d_out, &rec));
    BA_TRY(launch_geom_disp(ctx, nsrc, (const int32_t *)d_k, (const int64_t *)d_o, (const double *)d_p,
                            C, nobs, (const double *)d_e, (const double *)d_n, nu, (double *)d_out));
    BA_TRY(finish_out(ctx, &rec, 1));
    BA_HIP(hipStreamSynchronize(ctx->stream));   // `poff` is a stack temporary
    return BEATAMD_OK;
}

}  // extern "C"
