// ctx.hpp -- context, error handling, host/device argument staging, kernel timing.
// Internal to libbeat_amd.so (gfx950 only; no CUDA/portability layer).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/beat_amd.h"

namespace beatamd {

void set_error(const char *fmt, ...);

#define BA_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            beatamd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                               hipGetErrorString(e_));                                       \
            return BEATAMD_EHIP;                                                             \
        }                                                                                    \
    } while (0)

#define BA_CHECK(cond, code, ...)                                                            \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            beatamd::set_error(__VA_ARGS__);                                                 \
            return (code);                                                                   \
        }                                                                                    \
    } while (0)

#define BA_TRY(expr)                                                                         \
    do {                                                                                     \
        int rc_ = (expr);                                                                    \
        if (rc_ != BEATAMD_OK) return rc_;                                                   \
    } while (0)

// device status bits set by kernels (checked at synchronisation points)
enum : int { ST_INDEX_OOB = 1, ST_BAD_HYPO = 2, ST_NOT_PSD = 4, ST_BAD_COV = 8, ST_BAD_SCALE = 16 };

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
};

struct SeisLib {
    int64_t T = 0, P = 0, D = 0, S = 0, N = 0;
    double st_min = 0, st_dt = 1, du_min = 0, du_dt = 1;
    double *g = nullptr;  // HBM, (T,P,D,S,N) C-order, N fastest
    float *g32 = nullptr; // optional float copy of g (beatamd_seis_gflib_store_f32; g then holds float-representable values)
    bool owned = false;
    int64_t elems() const { return T * P * D * S * N; }
};

struct GeoLib {
    int64_t P = 0, Nobs = 0;
    double *g = nullptr;  // (P, Nobs)
};

struct WeightSet {
    int kind = BEATAMD_W_SCALAR;
    int64_t nd = 0, M = 0;
    double *w = nullptr;     // [nd] or [nd,M,M]
    double *slog = nullptr;  // [nd]
    int upper_tri = 0;       // dense only: exact zeros below the diagonal in every W
    // dense only: every W is upper-triangular AND banded -- no entry further than `band` columns right of the diagonal
    // exceeds 2^-40 of the largest entry of its matrix (-1: not banded / band > QF_BAND_MAX).  The whitening operator of
    // the reference's "exponential" noise structure (covariance.py:24-51: C_ij = exp(-|i-j| dt / t0), a Markov kernel) is
    // bidiagonal: band = 1.  wb [nd, M, band + 1]: row i = W[i, i .. i+band] (k_quadform_banded)
    int64_t band = -1;
    double *wb = nullptr;
    double dropped_rel = 0.0;   // banded: the largest entry beyond the band, relative to the largest of its row (<= 2^-40)
};

struct Laplacian {
    int64_t P = 0;
    double *L = nullptr;
    double logdet = 0;
};

struct Wavemap {
    std::vector<int32_t> libs;
    double *data = nullptr;  // [T,N]
    int32_t wset = -1;
    int64_t *hp_off = nullptr;     // device [T]
    int64_t *shift_off = nullptr;  // device [T] or nullptr
    // targets that share a station correction (the channels of a station: heart.py:2941-2950 repeats the station
    // indices per channel) have the same start times -> the index tables are built per SLOT = distinct shift variable
    int32_t nslot = 0;             // 0: no shifts, or every target has its own (tables per target)
    int32_t *tslot = nullptr;      // device [T]: slot of target t
    int64_t *slot_shift_off = nullptr;   // device [nslot]: the shift variable of the slot
    int interp = 0;
    int64_t T = 0, N = 0;
    bool f32 = false;   // read the libraries' float copies where a kernel supports it
};

struct Geodetic {
    std::vector<int32_t> libs;
    double *data = nullptr, *odws = nullptr;  // [Nobs]
    int64_t Nobs = 0;
    std::vector<int64_t> sizes;
    std::vector<int32_t> wsets;
    int64_t *hp_off = nullptr;  // device [nd]
};

// geometry-mode sources of the geodetic composite (analytic half space)
struct GeomSources {
    int nsrc = 0;
    int32_t *kind = nullptr;   // device [nsrc]
    int64_t *poff = nullptr;   // device [nsrc*10]
    double *pfix = nullptr;    // device [nsrc*10]
    double *east = nullptr, *north = nullptr, *los = nullptr;  // device [Nobs], [Nobs], [Nobs,3]
    int64_t Nobs = 0;
    double nu = 0.25;
};

struct FfiModel {
    beatamd_ffi_layout layout;
    int32_t nsub = 0;
    std::vector<int32_t> ndip, nstrike, patch_off;
    std::vector<double> patch_size;
    int64_t P = 0;
    int32_t *d_ndip = nullptr, *d_nstrike = nullptr, *d_patch_off = nullptr;
    double *d_patch_size = nullptr;
    std::vector<Wavemap> wavemaps;
    bool has_geo = false;
    Geodetic geo;
    bool geo_is_geometry = false;  // mu from analytic sources instead of G.T . slips
    GeomSources geom;
    int32_t lap = -1;
    int64_t nllk() const;
};

// A/B and test knobs of the stacking path (BEATAMD_G* / BEATAMD_WS_* environment variables; DESIGN.md 3.1b lists them).
// Read ONCE when the context is created (and again on beatamd_ctx_reload_knobs); a context created with
// BEATAMD_KNOBS_LIVE=1 in the environment -- the test suite, the A/B tools -- re-reads them at every stacking call so
// that one process can compare kernels.  KNOB_UNSET: the variable is not set, the default applies.
constexpr int KNOB_UNSET = -2147483647;
struct GfKnobs {
    int gf_kernel = KNOB_UNSET, gs_cg = KNOB_UNSET, gs_ws = KNOB_UNSET, gs_dma = KNOB_UNSET, gs_nt = KNOB_UNSET,
        ws_map = KNOB_UNSET, gs_pair = KNOB_UNSET, gs_nthint = KNOB_UNSET, gs_order = KNOB_UNSET, gs_fit = KNOB_UNSET,
        gs_win = KNOB_UNSET, gf_tinv = KNOB_UNSET, gs_tune = KNOB_UNSET, gf_order = KNOB_UNSET, gf_cgroup = KNOB_UNSET,
        gs_ml = KNOB_UNSET, gc_global = KNOB_UNSET, gc_sort = KNOB_UNSET, gc_keys = KNOB_UNSET, gc_bands = KNOB_UNSET, gr_cap = KNOB_UNSET,
        gr_pass_alloc = KNOB_UNSET, gr_var = KNOB_UNSET, sweep_v1 = KNOB_UNSET, qf_band = KNOB_UNSET, qf_fuse = KNOB_UNSET, gf_split = KNOB_UNSET, gm_wave = KNOB_UNSET;
    void read_env();
    static int get(int v, int dflt) { return v == KNOB_UNSET ? dflt : v; }
    static bool is(int v, int x) { return v != KNOB_UNSET && v == x; }       // set and equal to x
    static bool set(int v) { return v != KNOB_UNSET; }
};

struct KTimer {
    double total_ms = 0;
    int64_t n = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace beatamd

struct beatamd_ctx {
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    bool timing = false;
    std::map<std::string, beatamd::KTimer> timers;
    std::vector<hipEvent_t> event_pool;
    int *d_status = nullptr;  // device status word
    std::vector<beatamd::DevBuf> scratch;
    std::vector<std::unique_ptr<beatamd::SeisLib>> seislibs;
    std::vector<std::unique_ptr<beatamd::GeoLib>> geolibs;
    std::vector<std::unique_ptr<beatamd::WeightSet>> wsets;
    std::vector<std::unique_ptr<beatamd::Laplacian>> laps;
    std::vector<std::unique_ptr<beatamd::FfiModel>> models;
    int num_cu = 256;
    // name of the stacking kernel of the most recent launch (tests assert which kernel ran)
    char last_gf_kernel[96] = "";
    // distinct-row statistics of the most recent chain-shared launch (bench.py roofline leg)
    int64_t gs_ngtp = 0, gs_N = 0;
    double gs_trep = 1;             // targets served by one table cell (T / table slots)
    int gs_nvar = 1;                // slip variables: every distinct row is staged once per variable
    bool gs_has_passes = false;     // the statistics slot holds [rows per patch][passes per patch]
    // what the selection chose for the most recent stacking launch and why (beatamd_ctx_gf_plan)
    char gf_plan[384] = "";
    bool gf_band_fused = false;     // mode 3: the stacking kernel of the launch in hand evaluated the bidiagonal misfit itself
    char gf_tune_log[512] = "";   // the most recent group-size measurement (launch_gfstack), in words
    // largest distinct-row count of the previous small-group launch, read back asynchronously
    // (pinned mailbox + event; never waited for): sizes the next launch's row buffers
    uint32_t *h_umax = nullptr;
    hipEvent_t umax_event = nullptr;
    bool umax_pending = false;
    int umax_hist = -1, umax_hist_cg = 0;
    // measured chains-per-workgroup choice per problem shape: key -> (group size, row bound)
    std::map<std::vector<int64_t>, std::pair<int, int>> gs_tuned;
    int gs_cg = 0;
    beatamd::GfKnobs knobs;
    bool knobs_live = false;
    // device-resident Philox step counter (beatamd_ctx_set_step_counter): the proposal draws read it
    // instead of their `step` argument and advance it, so that a captured step replays correctly
    uint32_t *step_dev = nullptr;

    // grow-only scratch slot
    int get_scratch(int slot, size_t bytes, void **out);
    hipEvent_t get_event();
    void time_begin(const char *name);
    void time_end(const char *name);
    int check_status();  // sync + read status word; maps to BEATAMD_E*
};

namespace beatamd {

bool is_device_ptr(const void *p);
// the knobs in force for a stacking call (re-read from the environment first in live mode)
const GfKnobs &gf_knobs(beatamd_ctx *ctx);

// RAII-ish staging of one array argument.  Host pointers are mirrored in a scratch slot.
struct Arg {
    beatamd_ctx *ctx;
    void *host = nullptr;  // non-null => needs copy back (if out) after the launch
    void *dev = nullptr;
    size_t bytes = 0;
    bool out = false;
};

// in: returns device pointer holding the data (copy if host)
int stage_in(beatamd_ctx *ctx, int slot, const void *p, size_t bytes, const void **dev);
// out: returns device pointer to write into; if host, remember for copy_back
int stage_out(beatamd_ctx *ctx, int slot, void *p, size_t bytes, void **dev, Arg *rec,
              bool preload = false);
int finish_out(beatamd_ctx *ctx, Arg *recs, int n);  // D2H copies + sync if any host outs

struct ScopedTimer {
    beatamd_ctx *c;
    const char *n;
    ScopedTimer(beatamd_ctx *ctx, const char *name) : c(ctx), n(name) { c->time_begin(n); }
    ~ScopedTimer() { c->time_end(n); }
};

// scratch slot map (one per logical temporary so slots never alias within a call)
enum Slot : int {
    SL_IN0 = 0, SL_IN1, SL_IN2, SL_IN3, SL_IN4, SL_IN5, SL_IN6, SL_IN7,
    SL_OUT0, SL_OUT1, SL_OUT2,
    SL_ROWOFF, SL_WEIGHTS, SL_ST0, SL_RESID, SL_PARTIAL, SL_PARTIAL2, SL_QUAD, SL_MU, SL_SLIPS,
    SL_QPROP, SL_LPROP, SL_MISC, SL_GS_UROWS, SL_GS_UCOUNT, SL_GS_SLOT, SL_GS_W, SL_GS_UMAX, SL_GS_USLOT,
    SL_CHAINBAD, SL_Z, SL_ROWSCALE, SL_CUM, SL_STAGE2, SL_WHITEN,
    SL_CHOL_A, SL_CHOL_X, SL_CHOL_D, SL_CHOL_T, SL_CHOL_L,
    SL_GC_ORDER, SL_GS_ORDER, SL_GC_STREAM, SL_GC_HDR, SL_GC_META, SL_DELTA, SL_LOGU, SL_EDGES, SL_TSLOT, SL_SPLIT, SL_COUNT
};

}  // namespace beatamd
