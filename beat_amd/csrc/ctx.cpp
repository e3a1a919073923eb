// ctx.cpp -- context lifetime, staging, timing, error reporting.
#include "ctx.hpp"

#include <cstdlib>

namespace beatamd {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return BEATAMD_OK;
    // grow geometrically so that a sampler with slowly varying batch sizes settles
    size_t want = bytes + bytes / 4;
    if (p) {
        // kernels of earlier calls may still read the old buffer (any stream): drain first
        BA_HIP(hipDeviceSynchronize());
        BA_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&p, bytes);
        want = bytes;
    }
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        p = nullptr;
        return BEATAMD_ENOMEM;
    }
    cap = want;
    return BEATAMD_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int stage_in(beatamd_ctx *ctx, int slot, const void *p, size_t bytes, const void **dev)
{
    if (bytes == 0 || p == nullptr) {
        *dev = p;
        return BEATAMD_OK;
    }
    if (is_device_ptr(p)) {
        *dev = p;
        return BEATAMD_OK;
    }
    void *d = nullptr;
    BA_TRY(ctx->get_scratch(slot, bytes, &d));
    BA_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = d;
    return BEATAMD_OK;
}

int stage_out(beatamd_ctx *ctx, int slot, void *p, size_t bytes, void **dev, Arg *rec,
              bool preload)
{
    rec->ctx = ctx;
    rec->bytes = bytes;
    rec->out = true;
    if (bytes == 0 || p == nullptr || is_device_ptr(p)) {
        rec->host = nullptr;
        rec->dev = p;
        *dev = p;
        return BEATAMD_OK;
    }
    void *d = nullptr;
    BA_TRY(ctx->get_scratch(slot, bytes, &d));
    if (preload) BA_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    rec->host = p;
    rec->dev = d;
    *dev = d;
    return BEATAMD_OK;
}

int finish_out(beatamd_ctx *ctx, Arg *recs, int n)
{
    bool any = false;
    for (int i = 0; i < n; i++) {
        if (recs[i].host) {
            BA_HIP(hipMemcpyAsync(recs[i].host, recs[i].dev, recs[i].bytes, hipMemcpyDeviceToHost,
                                  ctx->stream));
            any = true;
        }
    }
    if (any) return ctx->check_status();  // synchronises
    return BEATAMD_OK;
}

int64_t FfiModel::nllk() const
{
    int64_t n = 1;  // like
    for (auto &w : wavemaps) n += w.T;
    if (has_geo) n += (int64_t)geo.sizes.size();
    if (lap >= 0) n += 1;
    return n;
}

}  // namespace beatamd

using namespace beatamd;

int beatamd_ctx::get_scratch(int slot, size_t bytes, void **out)
{
    if ((size_t)slot >= scratch.size()) scratch.resize(slot + 1);
    BA_TRY(scratch[slot].reserve(bytes));
    *out = scratch[slot].p;
    return BEATAMD_OK;
}

hipEvent_t beatamd_ctx::get_event()
{
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void beatamd_ctx::time_begin(const char *name)
{
    if (!timing) return;
    KTimer &t = timers[name];
    hipEvent_t a = get_event(), b = get_event();
    (void)hipEventRecord(a, stream);
    t.pending.emplace_back(a, b);
}

void beatamd_ctx::time_end(const char *name)
{
    if (!timing) return;
    KTimer &t = timers[name];
    if (!t.pending.empty()) (void)hipEventRecord(t.pending.back().second, stream);
}

int beatamd_ctx::check_status()
{
    BA_HIP(hipStreamSynchronize(stream));
    int st = 0;
    BA_HIP(hipMemcpy(&st, d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        BA_HIP(hipMemset(d_status, 0, sizeof(int)));
        if (st & ST_INDEX_OOB) {
            set_error("index out of bounds of the GF library "
                      "(duration/starttime outside the library grid)");
            return BEATAMD_EINDEX;
        }
        if (st & ST_BAD_HYPO) {
            set_error("nucleation index outside the patch grid");
            return BEATAMD_EINVAL;
        }
        if (st & ST_BAD_COV) {
            // smc.py:167-186: the text of the reference's ValueError
            set_error("Sample covariances contains Inf or NaN! Please try reducing the upper and lower bounds "
                      "of hyper parameters!");
            return BEATAMD_EBADCOV;
        }
        if (st & ST_NOT_PSD) {
            set_error("Matrix is not positive definite");   // numpy.linalg.LinAlgError's text
            return BEATAMD_ENOTPSD;
        }
        if (st & ST_BAD_SCALE) {
            set_error("PoissonProposal: a step width outside (0, 500] (or NaN): the draws of that parameter are NaN");
            return BEATAMD_EINVAL;
        }
    }
    return BEATAMD_OK;
}

namespace beatamd {

void GfKnobs::read_env()
{
    auto rd = [](const char *name) { const char *e = getenv(name); return e ? atoi(e) : KNOB_UNSET; };
    gf_kernel = rd("BEATAMD_GF_KERNEL"); gs_cg = rd("BEATAMD_GS_CG"); gs_ws = rd("BEATAMD_GS_WS"); gs_dma = rd("BEATAMD_GS_DMA");
    gs_nt = rd("BEATAMD_GS_NT"); ws_map = rd("BEATAMD_WS_MAP"); gs_pair = rd("BEATAMD_GS_PAIR"); gs_nthint = rd("BEATAMD_GS_NTHINT");
    gs_order = rd("BEATAMD_GS_ORDER"); gs_fit = rd("BEATAMD_GS_FIT"); gs_win = rd("BEATAMD_GS_WIN"); gf_tinv = rd("BEATAMD_GF_TINV");
    gs_tune = rd("BEATAMD_GS_TUNE"); gf_order = rd("BEATAMD_GF_ORDER"); gf_cgroup = rd("BEATAMD_GF_CGROUP"); gs_ml = rd("BEATAMD_GS_ML");
    gc_global = rd("BEATAMD_GC_GLOBAL"); gc_sort = rd("BEATAMD_GC_SORT"); gc_keys = rd("BEATAMD_GC_KEYS"); gc_bands = rd("BEATAMD_GC_BANDS"); gr_cap = rd("BEATAMD_GR_CAP");
    gr_pass_alloc = rd("BEATAMD_GR_PASS_ALLOC"); gr_var = rd("BEATAMD_GR_VAR"); sweep_v1 = rd("BEATAMD_SWEEP_V1"); qf_band = rd("BEATAMD_QF_BAND"); qf_fuse = rd("BEATAMD_QF_FUSE"); gf_split = rd("BEATAMD_GF_SPLIT"); gm_wave = rd("BEATAMD_GM_WAVE");
}

const GfKnobs &gf_knobs(beatamd_ctx *ctx)
{
    if (ctx->knobs_live) ctx->knobs.read_env();
    return ctx->knobs;
}

}  // namespace beatamd

extern "C" {

const char *beatamd_last_error(void) { return g_err; }

int beatamd_version(void) { return BEATAMD_VERSION; }

int beatamd_ctx_create(int device, beatamd_ctx **out)
{
    BA_CHECK(out != nullptr, BEATAMD_EINVAL, "ctx_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (%s)", hipGetErrorString(e));
        return BEATAMD_EHIP;
    }
    BA_CHECK(device >= 0 && device < ndev, BEATAMD_EINVAL, "device %d out of range [0,%d)", device,
             ndev);
    BA_HIP(hipSetDevice(device));
    beatamd_ctx *c = new beatamd_ctx();
    c->device = device;
    BA_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    BA_HIP(hipMalloc((void **)&c->d_status, sizeof(int)));
    BA_HIP(hipMemset(c->d_status, 0, sizeof(int)));
    hipDeviceProp_t prop;
    BA_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount;
    c->scratch.resize(SL_COUNT);
    c->knobs.read_env();
    {
        const char *live = getenv("BEATAMD_KNOBS_LIVE");
        c->knobs_live = live && atoi(live) != 0;
    }
    *out = c;
    return BEATAMD_OK;
}

int beatamd_ctx_reload_knobs(beatamd_ctx *c)
{
    BA_CHECK(c != nullptr, BEATAMD_EINVAL, "ctx_reload_knobs: NULL context");
    c->knobs.read_env();
    return BEATAMD_OK;
}

int beatamd_ctx_destroy(beatamd_ctx *c)
{
    if (!c) return BEATAMD_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &l : c->seislibs)
        if (l && l->owned && l->g) (void)hipFree(l->g);
    for (auto &l : c->seislibs)
        if (l && l->g32) (void)hipFree(l->g32);
    for (auto &l : c->geolibs)
        if (l && l->g) (void)hipFree(l->g);
    for (auto &w : c->wsets)
        if (w) {
            if (w->w) (void)hipFree(w->w);
            if (w->slog) (void)hipFree(w->slog);
        }
    for (auto &l : c->laps)
        if (l && l->L) (void)hipFree(l->L);
    for (auto &m : c->models)
        if (m) {
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
        }
    for (auto &s : c->scratch) s.release();
    for (auto &kv : c->timers)
        for (auto &pr : kv.second.pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->h_umax) (void)hipHostFree(c->h_umax);
    if (c->umax_event) (void)hipEventDestroy(c->umax_event);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return BEATAMD_OK;
}

int beatamd_ctx_set_stream(beatamd_ctx *c, void *s)
{
    BA_CHECK(c, BEATAMD_EINVAL, "ctx is NULL");
    BA_HIP(hipSetDevice(c->device));
    c->stream = (hipStream_t)s;  // NULL = the HIP null stream
    return BEATAMD_OK;
}

int beatamd_ctx_use_own_stream(beatamd_ctx *c)
{
    BA_CHECK(c, BEATAMD_EINVAL, "ctx is NULL");
    c->stream = c->own_stream;
    return BEATAMD_OK;
}

int beatamd_ctx_set_step_counter(beatamd_ctx *c, uint32_t *device_counter)
{
    BA_CHECK(c != nullptr, BEATAMD_EINVAL, "ctx is NULL");
    BA_CHECK(device_counter == nullptr || is_device_ptr(device_counter), BEATAMD_EINVAL,
             "ctx_set_step_counter: the counter must live in device memory");
    c->step_dev = device_counter;
    return BEATAMD_OK;
}

int beatamd_ctx_synchronize(beatamd_ctx *c)
{
    BA_CHECK(c, BEATAMD_EINVAL, "ctx is NULL");
    BA_HIP(hipSetDevice(c->device));
    return c->check_status();
}

int beatamd_ctx_enable_timing(beatamd_ctx *c, int on)
{
    BA_CHECK(c, BEATAMD_EINVAL, "ctx is NULL");
    c->timing = on != 0;
    return BEATAMD_OK;
}

static int drain_timer(beatamd_ctx *c, KTimer &t)
{
    for (auto &pr : t.pending) {
        BA_HIP(hipEventSynchronize(pr.second));
        float ms = 0.f;
        BA_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        t.total_ms += ms;
        t.n += 1;
        c->event_pool.push_back(pr.first);
        c->event_pool.push_back(pr.second);
    }
    t.pending.clear();
    return BEATAMD_OK;
}

int beatamd_ctx_kernel_time(beatamd_ctx *c, const char *kernel, double *total_ms, int64_t *launches)
{
    BA_CHECK(c && kernel, BEATAMD_EINVAL, "bad argument");
    auto it = c->timers.find(kernel);
    if (it == c->timers.end()) {
        if (total_ms) *total_ms = 0;
        if (launches) *launches = 0;
        return BEATAMD_OK;
    }
    BA_TRY(drain_timer(c, it->second));
    if (total_ms) *total_ms = it->second.total_ms;
    if (launches) *launches = it->second.n;
    return BEATAMD_OK;
}

int beatamd_ctx_reset_timing(beatamd_ctx *c)
{
    BA_CHECK(c, BEATAMD_EINVAL, "ctx is NULL");
    for (auto &kv : c->timers) {
        BA_TRY(drain_timer(c, kv.second));
        kv.second.total_ms = 0;
        kv.second.n = 0;
    }
    return BEATAMD_OK;
}

}  // extern "C"
