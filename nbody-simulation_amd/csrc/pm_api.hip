// C ABI of libpm.so: context management, host copies, stage sequencing.
#include "pm_internal.hpp"

#include <cstdlib>

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

thread_local std::string g_create_error;
std::once_flag g_fft_once;

int fail(pm_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

}  // namespace

void pm_set_create_error(const std::string &msg) { g_create_error = msg; }

namespace {

#define HIP_TRY(c, expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail((c), e_ == hipErrorOutOfMemory ? PM_ENOMEM : PM_EHIP,                \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

#define FFT_TRY(c, expr)                                                                     \
    do {                                                                                     \
        rocfft_status s_ = (expr);                                                           \
        if (s_ != rocfft_status_success)                                                     \
            return fail((c), PM_EFFT, std::string(#expr) + ": rocfft status " + std::to_string((int)s_)); \
    } while (0)

template <typename T>
int dalloc(pm_ctx *c, T **p, size_t count)
{
    if (*p) return PM_OK;
    HIP_TRY(c, hipMalloc((void **)p, count * sizeof(T) + 16));
    return PM_OK;
}

#define TRY(expr) do { int r_ = (expr); if (r_ != PM_OK) return r_; } while (0)

using stage_timer = pm_timer;
struct kernel_timer : pm_timer { kernel_timer(pm_ctx *c, int k) : pm_timer(c, PM_NSTAGES + k) {} };

// the event pairs of the stage and kernel timers, made when timing is first switched on
int make_timing_events(pm_ctx *c)
{
    for (int s = 0; s < pm_ctx::TIMED; ++s)
        for (int r = 0; r < pm_ctx::RING; ++r)
            for (int k = 0; k < 2; ++k)
                if (!c->ev[s][r][k]) HIP_TRY(c, hipEventCreate(&c->ev[s][r][k]));
    return PM_OK;
}

int bits_for(int64_t M)
{
    int b = 1;
    while (((int64_t)1 << b) < M) ++b;
    return b;
}

int alloc_particles(pm_ctx *c)
{
    size_t cap = (size_t)c->cap;
    for (int u = 0; u < 2; ++u) {
        TRY(dalloc(c, &c->pf[u], cap));
        if (c->pos_in_f64) TRY(dalloc(c, &c->pd[u], cap));
        if (c->vel_f64) TRY(dalloc(c, &c->vd[u], cap)); else TRY(dalloc(c, &c->vf[u], cap));
    }
    return PM_OK;
}

// The binning starts from the order before when the drift's estimate of the movers is at most PM_REBIN_SHARE_OF /
// PM_REBIN_SHARE of the particles.
#ifndef PM_REBIN_SHARE
#define PM_REBIN_SHARE 6
#endif
#ifndef PM_REBIN_SHARE_OF
#define PM_REBIN_SHARE_OF 1
#endif
// Buffers of the binning that starts from the cell order of the step before (every particle may be a mover).
int prepare_rebin(pm_ctx *c)
{
    if (c->rb_ctl) return PM_OK;
    const size_t cap = (size_t)c->cap;
    TRY(dalloc(c, &c->key_alt, cap + 4));
    HIP_TRY(c, hipMalloc((void **)&c->rb_ctl, pm_rebin_ctl_bytes(c->cap)));
    TRY(dalloc(c, &c->rb_before, cap / 64 + 2));
    for (int u = 0; u < 2; ++u) { TRY(dalloc(c, &c->rb_key[u], cap + 4)); TRY(dalloc(c, &c->rb_slot[u], cap + 4)); }
    TRY(dalloc(c, &c->rb_at, cap + 4));
    return PM_OK;
}

int ensure_binned(pm_ctx *c)
{
    if (c->binned) return PM_OK;
    stage_timer t(c, PM_STAGE_BIN);
    HIP_TRY(c, pm_settle_velocities(c));        // src is about to be overwritten
    if (!c->keys_fresh) HIP_TRY(c, pm_launch_cell_keys(c));
    TRY(pm_bin_order(c, bits_for(c->g.M)));
    HIP_TRY(c, pm_launch_reorder(c, !c->last_incremental));     // (with the row starts, unless the binning from the order before has brought them up to date itself)
    c->binned = true;
    c->order_known = true;
    c->order_n = c->n;
    // forces / forces64 are indexed by the storage slots of before the reorder: no longer attached to their particles
    c->have_forces = false; c->have_forces64 = false;
    return PM_OK;
}

int whole_mesh(pm_ctx *c)
{
    if (!c) return PM_EINVAL;
    if (c->slab) return fail(c, PM_ESTATE, "this call works on the whole mesh: a slab context takes the pm_slab_* calls");
    return PM_OK;
}

// the float32 mesh `which` must hold the output of the stage before, or what pm_set_mesh put there
int need_mesh(pm_ctx *c, int which, const char *what)
{
    // (a fused step computes the potential rows its own gather reads: whoever else wants the mesh gets the rest now)
    if (which == PM_MESH_POTENTIAL && (c->mesh_valid & (1u << which)) && c->potential_partial) return pm_complete_potential(c);
    if (c->mesh_valid & (1u << which)) return PM_OK;
    return fail(c, PM_ESTATE, std::string("no ") + what + ": run the stage that writes it (or pm_set_mesh) first; a fused pm_step keeps neither the density nor the gradient meshes");
}

int need_particles(pm_ctx *c)
{
    if (!c) return PM_EINVAL;
    if (!c->pf[0]) return fail(c, PM_ESTATE, "no particles: call pm_set_particles first");
    return PM_OK;
}

}  // namespace

// The drift leaves the new keys of the slots (key_in), a bit per slot that changed its key, and an estimate of how many
// did (the old keys, key_out, are still the sorted keys of this storage order).  Few: the new order is put together from
// the old one.  The estimate is a step or two old and decides speed only: the kernels count for themselves, and either
// way gives the same order.
int pm_bin_order(pm_ctx *c, int bits)
{
    const bool few = c->order_known && c->keys_fresh && c->movers_marked && c->movers_seen && c->n >= c->rebin_min && c->n < ((int64_t)1 << 30)
                     && !(c->flags & PM_FLAG_FULL_SORT) && (int64_t)*c->movers_seen * PM_REBIN_SHARE <= c->n * PM_REBIN_SHARE_OF;
    c->keys_fresh = false;      // the binning consumes key_in
    c->movers_marked = false;
    if (few) {
        TRY(prepare_rebin(c));
        HIP_TRY(c, pm_rebin_incremental(c, bits));
        c->last_incremental = true;
    } else {
        HIP_TRY(c, pm_sort_pairs(c, bits));
        c->last_incremental = false;
    }
    return PM_OK;
}

static int create_impl(pm_ctx *c)
{
    const int G = c->g.G;
    const size_t M = (size_t)c->g.M, cap = (size_t)c->cap;
    // a slab context holds the planes of its window only (and those of its slab, where the two may differ)
    const size_t plane = (size_t)G * G, planes = (size_t)(c->slab ? c->wmax : c->g.Gs);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, pm_noalloc_prepare(c->device));     // set-up for pm_energies_noalloc, which may not do it under a capture
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIP_TRY(c, hipHostMalloc((void **)&c->units_seen, 4 * sizeof(unsigned long long), hipHostMallocDefault));
    for (int k = 0; k < 4; ++k) c->units_seen[k] = 0ull;
    HIP_TRY(c, hipHostGetDevicePointer((void **)&c->units_seen_dev, (void *)c->units_seen, 0));
    {
        HIP_TRY(c, hipHostMalloc((void **)&c->movers_seen, sizeof(uint32_t), hipHostMallocDefault));
        *c->movers_seen = 0xffffffffu;      // nothing known yet: the first binnings sort the whole set
        HIP_TRY(c, hipHostGetDevicePointer((void **)&c->movers_seen_dev, (void *)c->movers_seen, 0));
        TRY(dalloc(c, &c->rb_mask, (size_t)c->cap / 64 + 2));
        TRY(dalloc(c, &c->mover_count, 1));
        HIP_TRY(c, hipMemset(c->mover_count, 0, sizeof(unsigned long long)));
    }
    if (c->flags & PM_FLAG_TIMING) TRY(make_timing_events(c));

    // the deposit reads keys and fraction records in groups of four: three entries of slack
    TRY(dalloc(c, &c->frac, cap + 4));
    TRY(dalloc(c, &c->key_in, cap + 4));
    TRY(dalloc(c, &c->key_out, cap + 4));
    TRY(dalloc(c, &c->src, cap));
    c->sort_tmp_bytes = pm_sort_tmp_bytes(c->cap);
    HIP_TRY(c, hipMalloc(&c->sort_tmp, c->sort_tmp_bytes + 16));
    TRY(dalloc(c, &c->row_start, planes * G + 1));
    TRY(dalloc(c, &c->heavy_rows, pm_heavy_row_entries(G, cap)));
    TRY(dalloc(c, &c->counters, 2 * 8256));        // two sets of the deposit's counters (pm_particles.hip: COUNTER_WORDS each, at most)
    HIP_TRY(c, hipMemsetAsync(c->counters, 0, 2 * 8256 * sizeof(uint32_t), c->stream));
    TRY(dalloc(c, &c->row_unit, planes * G + 1));
    if (!c->slab && c->fft_custom) TRY(dalloc(c, &c->row_live, plane / 64 + 2));
    if (!c->slab && c->fft_custom) TRY(dalloc(c, &c->row_need, plane / 64 + 2));
    TRY(dalloc(c, &c->unit_src, pm_max_units(cap)));
    TRY(dalloc(c, &c->unit_rows, pm_max_units(cap) * 4 * (size_t)G));
    TRY(dalloc(c, &c->light_rows, pm_heavy_row_entries(G, cap) * (size_t)G));
    // a slab context keeps one more density plane and three more potential planes than it owns, and no gradient meshes
    // (a context with room for windows always deposits z-transformed planes: no density mesh)
    if (!(c->slab && c->wmax != c->slabn)) TRY(dalloc(c, &c->density, (planes + 1) * plane));
    TRY(dalloc(c, &c->potential, (planes + 3) * plane));
    if (!c->slab)
        for (int d = 0; d < 3; ++d) TRY(dalloc(c, &c->grad[d], M));
    // (a slab's deposit may leave its nplanes + 1 planes as z-transformed rows: one more plane of spectrum)
    TRY(dalloc(c, &c->spectrum, (planes + (c->slab ? 1 : 0)) * G * c->g.Ghp));
    c->mesh_spec = c->spectrum; c->mesh_pot = c->potential + (c->slab ? plane : 0);
    if (c->slab && c->wmax != c->slabn) {
        c->mesh_spec = nullptr; c->mesh_pot = nullptr;
        TRY(dalloc(c, &c->mesh_spec, (size_t)c->slabn * G * c->g.Ghp));
        TRY(dalloc(c, &c->mesh_pot, (size_t)c->slabn * plane));
    }
    TRY(dalloc(c, &c->forces, 3 * cap));
    TRY(dalloc(c, &c->diag, 2 * 1024 + 4));
    HIP_TRY(c, hipMalloc(&c->stage, cap * 32 + 16));

    std::call_once(g_fft_once, [] { rocfft_setup(); });
    if (c->fft_custom) {
        HIP_TRY(c, pm_fft_upload_twiddles(c));
        return PM_OK;
    } else if (c->slab) {
        FFT_TRY(c, rocfft_execution_info_create(&c->info));      // the slab's own plans are made by pm_create_slab
        FFT_TRY(c, rocfft_execution_info_set_stream(c->info, c->stream));
        return PM_OK;
    } else {
        const size_t lengths[3] = {(size_t)G, (size_t)G, (size_t)G};
        FFT_TRY(c, rocfft_plan_create(&c->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                      rocfft_precision_single, 3, lengths, 1, nullptr));
        FFT_TRY(c, rocfft_plan_create(&c->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                      rocfft_precision_single, 3, lengths, 1, nullptr));
    }
    size_t wf = 0, wi = 0;
    FFT_TRY(c, rocfft_plan_get_work_buffer_size(c->fwd, &wf));
    FFT_TRY(c, rocfft_plan_get_work_buffer_size(c->inv, &wi));
    c->fft_work_bytes = wf > wi ? wf : wi;
    FFT_TRY(c, rocfft_execution_info_create(&c->info));
    if (c->fft_work_bytes) {
        HIP_TRY(c, hipMalloc(&c->fft_work, c->fft_work_bytes));
        FFT_TRY(c, rocfft_execution_info_set_work_buffer(c->info, c->fft_work, c->fft_work_bytes));
    }
    FFT_TRY(c, rocfft_execution_info_set_stream(c->info, c->stream));
    return PM_OK;
}

// A context for planes [plane0, plane0 + nplanes) of the mesh: all of them (pm_create) or a slab (pm_create_slab).
pm_ctx *pm_create_planes(int grid_size, double box_size, int64_t n_capacity, int device, unsigned flags, int plane0, int nplanes, bool slab, int window_planes)
{
    // cell keys are 32-bit: one key per cell of the planes this context's particles may lie in, and one more
    if (grid_size < 2 || nplanes < 1 || window_planes < nplanes || (int64_t)window_planes * grid_size * grid_size + 1 >= ((int64_t)1 << 32)) {
        fail(nullptr, PM_EINVAL, slab ? "the planes of a slab (or of its particle window) times grid_size^2 must be below 2^32 (cell keys are 32-bit)"
                                      : "grid_size must be in [2, 1625] (cell keys are 32-bit)");
        return nullptr;
    }
    if (!(box_size > 0.0) || !std::isfinite(box_size)) { fail(nullptr, PM_EINVAL, "box_size must be positive and finite"); return nullptr; }
    if (n_capacity < 0 || n_capacity >= ((int64_t)1 << 31)) { fail(nullptr, PM_EINVAL, "n_capacity must be in [0, 2^31-1]"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        fail(nullptr, PM_EHIP, "no such HIP device: " + std::to_string(device) + " of " + std::to_string(ndev));
        return nullptr;
    }
    if ((flags & PM_FLAG_F64) && ((flags & PM_FLAG_NGP) || slab)) {
        fail(nullptr, PM_EINVAL, "PM_FLAG_F64 is the cloud-in-cell branch on a whole mesh: not with PM_FLAG_NGP, not on a slab");
        return nullptr;
    }
    pm_ctx *c = new pm_ctx;
    c->g.G = grid_size; c->g.Gh = grid_size / 2 + 1;
    c->fft_custom = pm_fft_custom_supported(grid_size) && !(flags & PM_FLAG_ROCFFT_ONLY);
    c->g.Ghp = c->fft_custom ? pm_fft_pitch(grid_size) : c->g.Gh;
    c->g.M = (int64_t)grid_size * grid_size * grid_size;
    c->g.L = box_size; c->g.inv = (double)grid_size / box_size; c->g.Lf = (float)box_size;
    c->slab = slab;
    c->nslabs = grid_size / nplanes;
    c->g.i0 = plane0; c->g.Gs = nplanes; c->g.Gd = slab ? nplanes + 1 : grid_size; c->g.lo = slab ? 1 : 0;
    c->slab0 = plane0; c->slabn = nplanes; c->wmax = window_planes;
    c->g.sb = (c->fft_custom && !slab) ? pm_fft_spectrum_interleave(grid_size) : 0;
    c->device = device; c->flags = flags; c->cap = n_capacity > 0 ? n_capacity : 1;
    // (for tests: sets smaller than a million particles gain nothing from it, but every test of the path can then run on
    // the order put together from the order before)
    if (const char *v = std::getenv("PM_REBIN_MIN")) {
        char *end = nullptr;
        const long long m = std::strtoll(v, &end, 10);
        if (end != v && m >= 1) c->rebin_min = (int64_t)m;
    }
    if (create_impl(c) != PM_OK) {
        g_create_error = c->err;
        pm_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" {

const char *pm_last_error(const pm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// (a slab's leavers keep their slots until the next binning: they are not counted)
int64_t pm_num_particles(const pm_ctx *ctx) { return ctx ? ctx->n - ctx->leaving : -1; }

void pm_destroy(pm_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (int u = 0; u < 2; ++u) { hipFree(c->pf[u]); hipFree(c->pd[u]); hipFree(c->vf[u]); hipFree(c->vd[u]); }
    hipFree(c->frac); hipFree(c->elog);
    hipFree(c->key_in); hipFree(c->key_out); hipFree(c->src); hipFree(c->sort_tmp);
    hipFree(c->row_start); hipFree(c->heavy_rows); hipFree(c->counters); hipFree(c->unit_rows); hipFree(c->light_rows); hipFree(c->row_unit); hipFree(c->unit_src); hipFree(c->row_live); hipFree(c->row_need);
    if (c->slab && c->wmax != c->slabn) { hipFree(c->mesh_spec); hipFree(c->mesh_pot); }      // (else they alias spectrum / potential)
    hipFree(c->cuts);
    hipFree(c->density); hipFree(c->potential); hipFree(c->spectrum); hipFree(c->twiddle);
    for (int d = 0; d < 3; ++d) hipFree(c->grad[d]);
    hipFree(c->fft_work); hipFree(c->forces); hipFree(c->stage); hipFree(c->diag); hipFree(c->density64);
    hipFree(c->potential64); hipFree(c->spectrum64); hipFree(c->forces64); hipFree(c->fft_work64);
    for (int d = 0; d < 3; ++d) hipFree(c->grad64[d]);
    if (c->info64) rocfft_execution_info_destroy(c->info64);
    if (c->fwd64) rocfft_plan_destroy(c->fwd64);
    if (c->inv64) rocfft_plan_destroy(c->inv64);
    hipFree(c->outbox); hipFree(c->outbox_key); hipFree(c->outbox_key2); hipFree(c->outbox_idx2); hipFree(c->outbox_offsets);
    hipFree(c->outbox_count); hipFree(c->outbox_tmp);
    for (rocfft_plan p : {c->slab_fwd2, c->slab_inv2, c->slab_fwdx, c->slab_invx})
        if (p) rocfft_plan_destroy(p);
    if (c->info) rocfft_execution_info_destroy(c->info);
    if (c->fwd) rocfft_plan_destroy(c->fwd);
    if (c->inv) rocfft_plan_destroy(c->inv);
    for (int s = 0; s < pm_ctx::TIMED; ++s)
        for (int r = 0; r < pm_ctx::RING; ++r)
            for (int k = 0; k < 2; ++k)
                if (c->ev[s][r][k]) hipEventDestroy(c->ev[s][r][k]);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c->units_seen) hipHostFree((void *)c->units_seen);
    if (c->movers_seen) hipHostFree((void *)c->movers_seen);
    hipFree(c->mover_count); hipFree(c->key_alt); hipFree(c->rb_ctl); hipFree(c->rb_mask); hipFree(c->rb_before); hipFree(c->rb_at);
    for (int u = 0; u < 2; ++u) { hipFree(c->rb_key[u]); hipFree(c->rb_slot[u]); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->copy) { hipStreamSynchronize(c->copy); hipStreamDestroy(c->copy); }
    for (int b = 0; b < 2; ++b) {
        hipFree(c->snap[b]);
        if (c->snap_ready[b]) hipEventDestroy(c->snap_ready[b]);
        if (c->snap_done[b]) hipEventDestroy(c->snap_done[b]);
    }
    if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

pm_ctx *pm_create(int grid_size, double box_size, int64_t n_capacity, int device, unsigned flags)
{
    return pm_create_planes(grid_size, box_size, n_capacity, device, flags, 0, grid_size, false, grid_size);
}

int pm_set_stream(pm_ctx *c, void *hip_stream)
{
    if (!c) return PM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < pm_ctx::TIMED; ++s) pm_timer::flush(c, s);
    if (c->own_stream && c->stream) HIP_TRY(c, hipStreamDestroy(c->stream));
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    if (c->info) FFT_TRY(c, rocfft_execution_info_set_stream(c->info, c->stream));
    if (c->info64) FFT_TRY(c, rocfft_execution_info_set_stream(c->info64, c->stream));
    return PM_OK;
}

int pm_synchronize(pm_ctx *c)
{
    if (!c) return PM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PM_OK;
}

int pm_set_particles(pm_ctx *c, const void *pos, int pos_dtype, const void *vel, int vel_dtype, const float *mass, int64_t n)
{
    if (!c) return PM_EINVAL;
    if (n < 0 || n > c->cap) return fail(c, PM_EINVAL, "n outside [0, n_capacity]");
    if (!pos && n > 0) return fail(c, PM_EINVAL, "pos is NULL");
    if ((pos_dtype != PM_F32 && pos_dtype != PM_F64) || (vel_dtype != PM_F32 && vel_dtype != PM_F64))
        return fail(c, PM_EINVAL, "dtype must be PM_F32 or PM_F64");
    if ((c->flags & PM_FLAG_F64) && (pos_dtype != PM_F64 || vel_dtype != PM_F64))
        return fail(c, PM_EINVAL, "a PM_FLAG_F64 context takes float64 positions and velocities");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->pos_in_f64 = c->pos_f64 = (pos_dtype == PM_F64);
    c->vel_f64 = (vel_dtype == PM_F64);
    TRY(alloc_particles(c));
    // (a set large enough to be binned from the order before: its buffers now, not inside the first step that wants them)
    if (n >= c->rebin_min && !(c->flags & (PM_FLAG_FULL_SORT | PM_FLAG_F64 | PM_FLAG_NGP))) TRY(prepare_rebin(c));
    c->n = n; c->cur = 0; c->vcur = 0; c->v_stale = false; c->binned = false; c->have_forces = false; c->have_forces64 = false; c->keys_fresh = false; c->order_known = false;
    size_t ps = c->pos_f64 ? 8 : 4, vs = c->vel_f64 ? 8 : 4;
    if (n > 0) {
        // staging: positions (n,3), then the masses behind them
        char *st = (char *)c->stage;
        float *dmass = (float *)(st + (size_t)n * 24);
        HIP_TRY(c, hipMemcpyAsync(st, pos, (size_t)n * 3 * ps, hipMemcpyHostToDevice, c->stream));
        if (mass) HIP_TRY(c, hipMemcpyAsync(dmass, mass, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, pm_launch_pack(c, st, mass ? dmass : nullptr, nullptr));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (vel) HIP_TRY(c, hipMemcpyAsync(st, vel, (size_t)n * 3 * vs, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, pm_launch_pack(c, nullptr, nullptr, vel ? st : nullptr));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->deposits_of_set = 0;                             // a new set: crowded rows are planned for until a deposit has counted none
    return PM_OK;
}

static int check_generate(pm_ctx *c, int kind, int64_t n, const double *params, int nparams)
{
    const int want = pm_generate_param_count(kind);
    if (want < 0) return fail(c, PM_EINVAL, "kind must be PM_IC_UNIFORM, PM_IC_TWO_GALAXY or PM_IC_DISK");
    if (n < 0) return fail(c, PM_EINVAL, "n < 0");
    if (!params || nparams != want) return fail(c, PM_EINVAL, "this kind takes " + std::to_string(want) + " parameters");
    for (int k = 0; k < want; ++k)
        if (!std::isfinite(params[k])) return fail(c, PM_EINVAL, "parameter " + std::to_string(k) + " is not finite");
    // scale lengths and truncation radii must be positive (the rejection loops rely on it), dispersions non-negative
    const int npos = kind == PM_IC_TWO_GALAXY ? 3 : kind == PM_IC_DISK ? 4 : 0;
    for (int k = 0; k < npos; ++k)
        if (!(params[k] > 0.0)) return fail(c, PM_EINVAL, "parameter " + std::to_string(k) + " (a length) must be positive");
    if (params[want - 1] < 0.0) return fail(c, PM_EINVAL, "sigma_v must not be negative");
    return PM_OK;
}

int pm_generate(pm_ctx *c, int kind, int64_t n, uint64_t seed, const double *params, int nparams)
{
    if (!c) return PM_EINVAL;
    TRY(check_generate(c, kind, n, params, nparams));
    if (!c->slab && n > c->cap) return fail(c, PM_EINVAL, "n outside [0, n_capacity]");
    if (c->slab && n >= ((int64_t)1 << 32)) return fail(c, PM_EINVAL, "particle numbers travel as 32-bit ids: n must be below 2^32");
    if (c->slab && c->outbox_pending) return fail(c, PM_ESTATE, "the outbox of the last drift has not been taken yet");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->pos_in_f64 = c->pos_f64 = false;
    c->vel_f64 = false;
    TRY(alloc_particles(c));
    // (a set large enough to be binned from the order before: its buffers now, not inside the first step that wants them)
    if ((c->slab ? c->cap : n) >= c->rebin_min && !(c->flags & (PM_FLAG_FULL_SORT | PM_FLAG_F64 | PM_FLAG_NGP))) TRY(prepare_rebin(c));
    c->cur = 0; c->vcur = 0; c->v_stale = false; c->binned = false; c->have_forces = false; c->have_forces64 = false; c->keys_fresh = false; c->order_known = false;
    if (!c->slab) {
        c->n = n;
        HIP_TRY(c, pm_launch_generate(c, kind, n, seed, params));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->deposits_of_set = 0;
        return PM_OK;
    }
    // a slab: every rank draws all n numbers and keeps the particles of its window, in the order of their numbers --
    // how many each block of 256 numbers keeps, the running sum of that (on the host: a few megabytes, once), the records
    c->n = 0; c->leaving = 0; c->slab_check = false;
    const size_t blocks = (size_t)((n + 255) / 256);
    if (blocks == 0) return PM_OK;
    uint32_t *dcount = nullptr;
    HIP_TRY(c, hipMalloc((void **)&dcount, blocks * sizeof(uint32_t)));
    std::vector<uint32_t> h(blocks);
    int rc = PM_OK;
    hipError_t e = pm_launch_generate_count(c, kind, n, seed, params, dcount);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), dcount, blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    uint64_t total = 0;
    if (e == hipSuccess) {
        for (size_t b = 0; b < blocks; ++b) { const uint32_t k = h[b]; h[b] = (uint32_t)total; total += k; }
        if ((int64_t)total > c->cap)
            rc = fail(c, PM_ENOMEM, std::to_string(total) + " particles of the set lie in this slab's window, its capacity is " + std::to_string(c->cap));
    }
    if (e == hipSuccess && rc == PM_OK) e = hipMemcpyAsync(dcount, h.data(), blocks * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && rc == PM_OK) e = pm_launch_generate_keep(c, kind, n, seed, params, dcount);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dcount);
    if (e != hipSuccess) return fail(c, PM_EHIP, std::string("pm_generate: ") + hipGetErrorString(e));
    if (rc != PM_OK) return rc;
    c->n = (int64_t)total;
    HIP_TRY(c, hipMemsetAsync(c->outbox_count, 0, (size_t)(2 + c->nslabs) * sizeof(uint32_t), c->stream));
    return PM_OK;
}

int pm_generate_plane_counts(int device, int grid_size, double box_size, int kind, int64_t n, uint64_t seed, const double *params, int nparams,
                             int64_t *host_counts)
{
    if (grid_size < 2 || grid_size > 16384 || !(box_size > 0.0) || !std::isfinite(box_size)) return fail(nullptr, PM_EINVAL, "grid_size must be in [2, 16384] and box_size positive");
    TRY(check_generate(nullptr, kind, n, params, nparams));
    if (!host_counts) return fail(nullptr, PM_EINVAL, "host_counts is NULL");
    HIP_TRY(nullptr, hipSetDevice(device));
    pm_geom g{};
    g.G = grid_size; g.L = box_size; g.inv = (double)grid_size / box_size; g.Lf = (float)box_size; g.Gs = grid_size; g.Gd = grid_size;
    const size_t G = (size_t)grid_size;
    unsigned long long *d = nullptr;
    HIP_TRY(nullptr, hipMalloc((void **)&d, G * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, G * sizeof(unsigned long long));
    if (e == hipSuccess) e = pm_launch_generate_planes(g, nullptr, kind, n, seed, params, d);
    static_assert(sizeof(unsigned long long) == sizeof(int64_t), "the counters are copied out as they are");
    if (e == hipSuccess) e = hipMemcpy(host_counts, d, G * sizeof(int64_t), hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(nullptr, PM_EHIP, std::string("pm_generate_plane_counts: ") + hipGetErrorString(e));
    return PM_OK;
}

static int deposit_impl(pm_ctx *c, bool to_spectrum)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(ensure_binned(c));
    stage_timer t(c, PM_STAGE_DEPOSIT);
    HIP_TRY(c, pm_launch_deposit(c, to_spectrum));
    if (to_spectrum) c->mesh_valid &= ~(1u << PM_MESH_DENSITY);     // the rows left as half-spectrum rows
    else c->mesh_valid |= 1u << PM_MESH_DENSITY;
    return PM_OK;
}

int pm_deposit(pm_ctx *c) { return deposit_impl(c, false); }

// have_z: the half spectrum already holds the z-transformed density rows (fused deposit)
// needed_rows (a fused step, right after its deposit): the last pass, inverse along z, is run for the rows that the step's
// gather reads only (pm_ctx::row_need); the potential mesh is then partial until someone asks for all of it
static int potential_impl(pm_ctx *c, bool have_z, bool needed_rows = false)
{
    TRY(whole_mesh(c));
    if (!c) return PM_EINVAL;
    if (!have_z) TRY(need_mesh(c, PM_MESH_DENSITY, "density"));
    HIP_TRY(c, hipSetDevice(c->device));
    c->mesh_valid |= 1u << PM_MESH_POTENTIAL;
#ifdef PM_POTENTIAL_WHOLE
    needed_rows = false;
#endif
    needed_rows = needed_rows && have_z && c->fft_custom && c->rows_marked && !c->want_full_potential;
    c->potential_partial = needed_rows;
    if (c->fft_custom && !(c->flags & PM_FLAG_TIMING)) {
        // The three column passes (along y, along x with the Green multiply, back along y) touch the same columns c:
        // they are run over one group of column tiles after the other, a group being small enough (200 MB) that
        // much of what one pass wrote is still in the last-level cache when the next pass reads it.  With stage
        // timing on, the passes run over the whole spectrum one after the other instead (below), so that each can
        // be timed.
        if (!have_z) HIP_TRY(c, pm_launch_row_fft(c, true));
        const int tiles = pm_fft_tiles(c);
        // (measured: 6 tiles of 16 columns at G = 512 gain 1.5 % of the step; at G = 1024, where a tile is 134 MB and
        // a launch fills the GPU only once, any grouping loses to its launch tails)
#ifndef PM_COL_GROUP
#define PM_COL_GROUP 6
#endif
        const int group = c->g.G <= 512 ? PM_COL_GROUP : tiles;
        for (int t0 = 0; t0 < tiles; t0 += group) {
            const int nt = t0 + group <= tiles ? group : tiles - t0;
            HIP_TRY(c, pm_launch_col_fft_tiles(c, -1, t0, nt, have_z && c->rows_marked));
            HIP_TRY(c, pm_launch_col_fft_tiles(c, 0, t0, nt));
            HIP_TRY(c, pm_launch_col_fft_tiles(c, 1, t0, nt));
        }
        HIP_TRY(c, pm_launch_row_fft(c, false, needed_rows));
        return PM_OK;
    }
    {
        stage_timer t(c, PM_STAGE_FFT_FWD);
        if (c->fft_custom) {
            if (!have_z) HIP_TRY(c, pm_launch_row_fft(c, true));
            kernel_timer k(c, PM_KERNEL_COL_Y_FWD);
            HIP_TRY(c, pm_launch_col_fft(c, -1, have_z && c->rows_marked));
        } else {
            void *in[1] = {c->density}, *out[1] = {c->spectrum};
            FFT_TRY(c, rocfft_execute(c->fwd, in, out, c->info));
        }
    }
    {
        stage_timer t(c, PM_STAGE_GREEN);
        if (c->fft_custom) {
            kernel_timer k(c, PM_KERNEL_COL_X_GREEN);
            HIP_TRY(c, pm_launch_col_fft(c, 0));
        } else HIP_TRY(c, pm_launch_green(c));
    }
    {
        stage_timer t(c, PM_STAGE_FFT_INV);
        if (c->fft_custom) {
            {
                kernel_timer k(c, PM_KERNEL_COL_Y_INV);
                HIP_TRY(c, pm_launch_col_fft(c, 1));
            }
            kernel_timer k(c, PM_KERNEL_ROW_INV);
            HIP_TRY(c, pm_launch_row_fft(c, false, needed_rows));
        } else {
            void *in[1] = {c->spectrum}, *out[1] = {c->potential};
            FFT_TRY(c, rocfft_execute(c->inv, in, out, c->info));
        }
    }
    return PM_OK;
}

int pm_potential(pm_ctx *c) { return potential_impl(c, false); }

// The rows of the potential mesh that the last fused step left out (potential_impl), from the half spectrum, which holds the
// input of the inverse transform along z until the next potential is computed.  No allocation, nothing waited for.
extern "C++" int pm_complete_potential(pm_ctx *c)
{
    if (!c->potential_partial) return PM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, pm_launch_row_fft(c, false, false));
    c->potential_partial = false;
    return PM_OK;
}

int pm_set_softening(pm_ctx *c, double length)
{
    if (!c) return PM_EINVAL;
    if (!(length >= 0.0) || !std::isfinite(length)) return fail(c, PM_EINVAL, "softening length must be finite and not negative");
    // exp(-(sigma k_phys)^2 / 2) with k_phys = 2 pi k G / L, as a coefficient of k^2 in (cycles/sample)^2
    const double a = 2.0 * 3.14159265358979323846 * length * (double)c->g.G / c->g.L;
    c->g.soft = (float)(0.5 * a * a);
    return PM_OK;
}

int pm_gradient(pm_ctx *c)
{
    TRY(whole_mesh(c));
    if (!c) return PM_EINVAL;
    TRY(need_mesh(c, PM_MESH_POTENTIAL, "potential"));
    HIP_TRY(c, hipSetDevice(c->device));
    stage_timer t(c, PM_STAGE_GRADIENT);
    HIP_TRY(c, pm_launch_gradient(c));
    c->mesh_valid |= (1u << PM_MESH_GX) | (1u << PM_MESH_GY) | (1u << PM_MESH_GZ);
    return PM_OK;
}

int pm_forces(pm_ctx *c)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    for (int w = PM_MESH_GX; w <= PM_MESH_GZ; ++w) TRY(need_mesh(c, w, "gradient"));
    HIP_TRY(c, hipSetDevice(c->device));
    stage_timer t(c, PM_STAGE_GATHER);
    HIP_TRY(c, pm_launch_gather(c, 0.0, false, true, false));
    c->have_forces = true;
    return PM_OK;
}

static int gather_kick_drift(pm_ctx *c, double dt, bool fused)
{
    stage_timer t(c, PM_STAGE_GATHER);
    kernel_timer k(c, PM_KERNEL_GATHER);
    HIP_TRY(c, pm_launch_gather(c, dt, true, false, fused));
    c->pos_f64 = false;      // positions are float32 from the first drift on (main.py:60)
    c->binned = false;
    c->have_forces = false;
    c->keys_fresh = true;
    return PM_OK;
}

int pm_gather_kick_drift(pm_ctx *c, double dt)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    for (int w = PM_MESH_GX; w <= PM_MESH_GZ; ++w) TRY(need_mesh(c, w, "gradient"));
    HIP_TRY(c, hipSetDevice(c->device));
    return gather_kick_drift(c, dt, false);
}

static int step_ngp(pm_ctx *c, double dt);
static int step_f64(pm_ctx *c, double dt);

int pm_step(pm_ctx *c, double dt, int nsteps)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    if (nsteps < 0) return fail(c, PM_EINVAL, "nsteps < 0");
    if (c->flags & PM_FLAG_NGP) {
        for (int s = 0; s < nsteps; ++s) TRY(step_ngp(c, dt));
        return PM_OK;
    }
    if (c->flags & PM_FLAG_F64) {
        for (int s = 0; s < nsteps; ++s) TRY(step_f64(c, dt));
        return PM_OK;
    }
    const bool fused = (c->flags & PM_FLAG_FUSED_GRADIENT) != 0;
    for (int s = 0; s < nsteps; ++s) {
        // fused mode on the hand-written FFT path: the deposit leaves z-transformed rows, no density mesh
        const bool z_in_deposit = fused && c->fft_custom;
        TRY(deposit_impl(c, z_in_deposit));
        TRY(potential_impl(c, z_in_deposit, fused));
        if (!fused) TRY(pm_gradient(c));
        else c->mesh_valid &= ~((1u << PM_MESH_GX) | (1u << PM_MESH_GY) | (1u << PM_MESH_GZ));     // differenced on the fly
        TRY(gather_kick_drift(c, dt, fused));
    }
    return PM_OK;
}

// main.py:44-67 with the trajectory copies beside the steps
int pm_run(pm_ctx *c, double dt, int nsteps, float *trajectory, int energy_every, double *ke, double *pe)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    if (nsteps < 0 || energy_every < 0) return fail(c, PM_EINVAL, "nsteps < 0 or energy_every < 0");
    if (nsteps == 0) return PM_OK;
    if (energy_every > 0 && (!ke || !pe)) return fail(c, PM_EINVAL, "energy_every > 0 needs ke and pe");
    // (the snapshots are the float32 positions of main.py:60-61; a PM_FLAG_F64 context keeps float64 positions)
    if (trajectory && (c->flags & PM_FLAG_F64)) return fail(c, PM_ESTATE, "pm_run records float32 trajectories: not on a PM_FLAG_F64 context (step and read the positions instead)");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->n, snap_bytes = n * 3 * sizeof(float);
    const int nsamples = energy_every > 0 ? (nsteps - 1) / energy_every + 1 : 0;
    if (trajectory && n) {
        if (!c->copy) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            if (!c->snap[b]) HIP_TRY(c, hipMalloc((void **)&c->snap[b], (size_t)c->cap * 3 * sizeof(float)));
            if (!c->snap_ready[b]) HIP_TRY(c, hipEventCreateWithFlags(&c->snap_ready[b], hipEventDisableTiming));
            if (!c->snap_done[b]) HIP_TRY(c, hipEventCreateWithFlags(&c->snap_done[b], hipEventDisableTiming));
        }
    }
    if ((size_t)nsamples > c->elog_cap) {   // (nsamples, 2) on the device: grown when a run samples more than any before it
        if (c->elog) { hipFree(c->elog); c->elog = nullptr; c->elog_cap = 0; }
        HIP_TRY(c, hipMalloc((void **)&c->elog, (size_t)nsamples * 2 * sizeof(double)));
        c->elog_cap = (size_t)nsamples;
    }
    double *elog = c->elog;
    int rc = PM_OK;
    for (int s = 0; s < nsteps && rc == PM_OK; ++s) {
        c->want_full_potential = energy_every > 0 && s % energy_every == 0;    // (the sample reads the mesh at the new positions)
        rc = pm_step(c, dt, 1);
        c->want_full_potential = false;
        if (rc != PM_OK) break;
        if (trajectory && n) {
            const int b = s & 1;
            hipError_t e = hipSuccess;
            // the copy that read this buffer two steps ago must have finished
            if (s >= 2) e = hipStreamWaitEvent(c->stream, c->snap_done[b], 0);
            // main.py:60: positions are float32 from the first drift on
            if (e == hipSuccess) e = pm_launch_export_particles(c, c->snap[b], nullptr);
            if (e == hipSuccess) e = hipEventRecord(c->snap_ready[b], c->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->copy, c->snap_ready[b], 0);
            if (e == hipSuccess) e = hipMemcpyAsync(trajectory + (size_t)s * n * 3, c->snap[b], snap_bytes, hipMemcpyDeviceToHost, c->copy);
            if (e == hipSuccess) e = hipEventRecord(c->snap_done[b], c->copy);
            if (e != hipSuccess) { rc = fail(c, PM_EHIP, hipGetErrorString(e)); break; }
        }
        if (nsamples && s % energy_every == 0) {
            const int k = s / energy_every;
            hipError_t e = pm_launch_energies(c, elog + 2 * k, elog + 2 * k + 1);
            if (e != hipSuccess) { rc = fail(c, PM_EHIP, hipGetErrorString(e)); break; }
        }
    }
    // everything queued so far is drained also after an error: the caller's arrays must not be written to later
    hipError_t e1 = c->copy ? hipStreamSynchronize(c->copy) : hipSuccess;
    hipError_t e2 = hipStreamSynchronize(c->stream);
    if (rc == PM_OK && nsamples) {
        std::vector<double> host((size_t)nsamples * 2);
        hipError_t e = hipMemcpy(host.data(), elog, host.size() * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, PM_EHIP, hipGetErrorString(e));
        else for (int k = 0; k < nsamples; ++k) { ke[k] = host[2 * k]; pe[k] = host[2 * k + 1]; }
    }
    if (rc == PM_OK && e1 != hipSuccess) rc = fail(c, PM_EHIP, hipGetErrorString(e1));
    if (rc == PM_OK && e2 != hipSuccess) rc = fail(c, PM_EHIP, hipGetErrorString(e2));
    return rc;
}

void *pm_host_alloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { pm_set_create_error(std::string("pm_host_alloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}

void pm_host_free(void *p)
{
    if (p) hipHostFree(p);
}

int pm_get_particles(pm_ctx *c, void *pos, int *pos_dtype, void *vel, int *vel_dtype)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->n;
    if (pos_dtype) *pos_dtype = c->pos_f64 ? PM_F64 : PM_F32;
    if (vel_dtype) *vel_dtype = c->vel_f64 ? PM_F64 : PM_F32;
    if (pos && n) {
        HIP_TRY(c, pm_launch_export_particles(c, c->stage, nullptr));
        HIP_TRY(c, hipMemcpyAsync(pos, c->stage, n * 3 * (c->pos_f64 ? 8 : 4), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (vel && n) {
        HIP_TRY(c, pm_launch_export_particles(c, nullptr, c->stage));
        HIP_TRY(c, hipMemcpyAsync(vel, c->stage, n * 3 * (c->vel_f64 ? 8 : 4), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return PM_OK;
}

int pm_get_forces(pm_ctx *c, float *forces)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    if (!c->have_forces) return fail(c, PM_ESTATE, "no forces: call pm_forces first");
    if (!forces) return fail(c, PM_EINVAL, "forces is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n) {
        HIP_TRY(c, pm_launch_export_forces(c, (float *)c->stage));
        HIP_TRY(c, hipMemcpyAsync(forces, c->stage, (size_t)c->n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return PM_OK;
}

static float *mesh_ptr(pm_ctx *c, int which)
{
    switch (which) {
    case PM_MESH_DENSITY: return c->density;
    case PM_MESH_POTENTIAL: return c->potential;
    case PM_MESH_GX: return c->grad[0];
    case PM_MESH_GY: return c->grad[1];
    case PM_MESH_GZ: return c->grad[2];
    default: return nullptr;
    }
}

int pm_get_mesh(pm_ctx *c, int which, float *out)
{
    if (!c) return PM_EINVAL;
    if (c->slab) return fail(c, PM_ESTATE, "a slab context has no whole meshes: use pm_slab_planes");
    float *p = mesh_ptr(c, which);
    if (!p || !out) return fail(c, PM_EINVAL, "bad mesh selector or NULL buffer");
    TRY(need_mesh(c, which, which == PM_MESH_DENSITY ? "density" : which == PM_MESH_POTENTIAL ? "potential" : "gradient"));
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, p, (size_t)c->g.M * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PM_OK;
}

int pm_set_mesh(pm_ctx *c, int which, const float *in)
{
    if (!c) return PM_EINVAL;
    if (c->slab) return fail(c, PM_ESTATE, "a slab context has no whole meshes: use pm_slab_planes");
    float *p = mesh_ptr(c, which);
    if (!p || !in) return fail(c, PM_EINVAL, "bad mesh selector or NULL buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(p, in, (size_t)c->g.M * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->mesh_valid |= 1u << which;
    if (which == PM_MESH_POTENTIAL) c->potential_partial = false;
    return PM_OK;
}

int pm_get_cell_keys(pm_ctx *c, int64_t *keys, int64_t *perm)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    if (!keys || !perm) return fail(c, PM_EINVAL, "NULL buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(ensure_binned(c));
    if (c->n) {
        // both exports fit the staging buffer: 2 * n * 8 <= cap * 24 bytes
        int64_t *dk = (int64_t *)c->stage, *dp = dk + c->n;
        HIP_TRY(c, pm_launch_export_keys(c, dk, dp));
        HIP_TRY(c, hipMemcpyAsync(keys, dk, (size_t)c->n * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(perm, dp, (size_t)c->n * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return PM_OK;
}

int pm_last_binning(pm_ctx *c, int64_t *movers)
{
    // (the results are 0 and 1: errors are reported as negated codes)
    if (!c) return -PM_EINVAL;
    if (!movers) { fail(c, PM_EINVAL, "movers is NULL"); return -PM_EINVAL; }
    *movers = 0;
    if (!c->last_incremental) {
        // the drift's estimate, as the host last saw it (what the choice was made by, or a newer one)
        const uint32_t seen = c->movers_seen ? *c->movers_seen : 0xffffffffu;
        *movers = seen == 0xffffffffu ? -1 : (int64_t)seen;
        return 0;
    }
    uint32_t ctl[PM_RB_WORDS];
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMemcpyAsync(ctl, c->rb_ctl, sizeof(ctl), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { fail(c, PM_EHIP, std::string("pm_last_binning: ") + hipGetErrorString(e)); return -PM_EHIP; }
    *movers = (int64_t)ctl[PM_RB_MOVERS];
    return 1;
}

int pm_energies(pm_ctx *c, double *ke, double *pe)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    if (!ke || !pe) return fail(c, PM_EINVAL, "NULL buffer");
    if ((c->flags & (PM_FLAG_NGP | PM_FLAG_F64)) && !c->potential64) return fail(c, PM_ESTATE, "no float64 potential yet: step, or call pm_potential_f64 / pm_set_mesh_f64 first");
    if (!(c->flags & (PM_FLAG_NGP | PM_FLAG_F64))) TRY(need_mesh(c, PM_MESH_POTENTIAL, "potential"));
    HIP_TRY(c, hipSetDevice(c->device));
    double *res = c->diag + 2 * 1024;
    HIP_TRY(c, pm_launch_energies(c, res, res + 1));
    double host[2] = {0.0, 0.0};
    HIP_TRY(c, hipMemcpyAsync(host, res, sizeof(host), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *ke = host[0]; *pe = host[1];
    return PM_OK;
}

int pm_forces_ngp(pm_ctx *c)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    for (int w = PM_MESH_GX; w <= PM_MESH_GZ; ++w) TRY(need_mesh(c, w, "gradient"));
    HIP_TRY(c, hipSetDevice(c->device));
    stage_timer t(c, PM_STAGE_GATHER);
    HIP_TRY(c, pm_launch_force_ngp(c));
    c->have_forces = true;
    return PM_OK;
}

int pm_deposit_ngp(pm_ctx *c)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(dalloc(c, &c->density64, (size_t)c->g.M));
    // the sort scratch (keys, src) is shared with the cloud-in-cell binning: settle what depends on it
    HIP_TRY(c, pm_settle_velocities(c));
    c->binned = false; c->keys_fresh = false; c->order_known = false;
    uint32_t *bad = (uint32_t *)(c->diag + 2 * 1024 + 2);
    HIP_TRY(c, hipMemsetAsync(bad, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->density64, 0, (size_t)c->g.M * sizeof(double), c->stream));
    HIP_TRY(c, pm_launch_ngp_keys(c, bad));
    HIP_TRY(c, pm_sort_pairs(c, bits_for(c->g.M)));
    HIP_TRY(c, pm_launch_ngp_sum(c));
    uint32_t h = 0;
    HIP_TRY(c, hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h) return fail(c, PM_ERANGE, "NGP index out of range: a position wraps to exactly box_size (the reference raises IndexError here)");
    return PM_OK;
}

// ---- the NGP branch in float64 ----------------------------------------------------------------------

// meshes, half spectrum and 3-D rocFFT plans in double precision, on first use
static int ensure_f64(pm_ctx *c)
{
    const size_t M = (size_t)c->g.M;
    const int G = c->g.G;
    TRY(dalloc(c, &c->density64, M));
    TRY(dalloc(c, &c->potential64, M));
    for (int d = 0; d < 3; ++d) TRY(dalloc(c, &c->grad64[d], M));
    TRY(dalloc(c, &c->spectrum64, (size_t)G * G * c->g.Gh));
    TRY(dalloc(c, &c->forces64, 3 * (size_t)c->cap));
    if (!c->fwd64) {
        const size_t lengths[3] = {(size_t)G, (size_t)G, (size_t)G};
        FFT_TRY(c, rocfft_plan_create(&c->fwd64, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                      rocfft_precision_double, 3, lengths, 1, nullptr));
        FFT_TRY(c, rocfft_plan_create(&c->inv64, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                      rocfft_precision_double, 3, lengths, 1, nullptr));
        size_t wf = 0, wi = 0;
        FFT_TRY(c, rocfft_plan_get_work_buffer_size(c->fwd64, &wf));
        FFT_TRY(c, rocfft_plan_get_work_buffer_size(c->inv64, &wi));
        const size_t work = wf > wi ? wf : wi;
        FFT_TRY(c, rocfft_execution_info_create(&c->info64));
        if (work) {
            HIP_TRY(c, hipMalloc(&c->fft_work64, work));
            FFT_TRY(c, rocfft_execution_info_set_work_buffer(c->info64, c->fft_work64, work));
        }
    }
    FFT_TRY(c, rocfft_execution_info_set_stream(c->info64, c->stream));
    return PM_OK;
}

static double *mesh64_ptr(pm_ctx *c, int which)
{
    switch (which) {
    case PM_MESH_DENSITY: return c->density64;
    case PM_MESH_POTENTIAL: return c->potential64;
    case PM_MESH_GX: return c->grad64[0];
    case PM_MESH_GY: return c->grad64[1];
    case PM_MESH_GZ: return c->grad64[2];
    default: return nullptr;
    }
}

int pm_get_mesh_f64(pm_ctx *c, int which, double *out)
{
    TRY(whole_mesh(c));
    double *p = mesh64_ptr(c, which);
    if (which < PM_MESH_DENSITY || which > PM_MESH_GZ || !out) return fail(c, PM_EINVAL, "bad mesh selector or NULL buffer");
    if (!p) return fail(c, PM_ESTATE, "no float64 meshes yet: call pm_deposit_ngp or pm_set_mesh_f64 first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, p, (size_t)c->g.M * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PM_OK;
}

int pm_set_mesh_f64(pm_ctx *c, int which, const double *in)
{
    TRY(whole_mesh(c));
    if (which < PM_MESH_DENSITY || which > PM_MESH_GZ || !in) return fail(c, PM_EINVAL, "bad mesh selector or NULL buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(ensure_f64(c));
    HIP_TRY(c, hipMemcpyAsync(mesh64_ptr(c, which), in, (size_t)c->g.M * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PM_OK;
}

// functions.py:73-89 on a float64 density.  The mean that :81-82 subtract only moves the DC term, which :86 zeroes.
int pm_potential_f64(pm_ctx *c)
{
    TRY(whole_mesh(c));
    if (!c->density64) return fail(c, PM_ESTATE, "no float64 density: call pm_deposit_ngp or pm_set_mesh_f64 first");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(ensure_f64(c));
    {
        stage_timer t(c, PM_STAGE_FFT_FWD);
        void *in[1] = {c->density64}, *out[1] = {c->spectrum64};
        FFT_TRY(c, rocfft_execute(c->fwd64, in, out, c->info64));
    }
    {
        stage_timer t(c, PM_STAGE_GREEN);
        HIP_TRY(c, pm_launch_green64(c));
    }
    {
        stage_timer t(c, PM_STAGE_FFT_INV);
        void *in[1] = {c->spectrum64}, *out[1] = {c->potential64};
        FFT_TRY(c, rocfft_execute(c->inv64, in, out, c->info64));
    }
    return PM_OK;
}

int pm_gradient_f64(pm_ctx *c)
{
    TRY(whole_mesh(c));
    if (!c->potential64) return fail(c, PM_ESTATE, "no float64 potential: call pm_potential_f64 or pm_set_mesh_f64 first");
    HIP_TRY(c, hipSetDevice(c->device));
    stage_timer t(c, PM_STAGE_GRADIENT);
    HIP_TRY(c, pm_launch_gradient64(c));
    return PM_OK;
}

int pm_forces_ngp_f64(pm_ctx *c)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    if (!c->grad64[0]) return fail(c, PM_ESTATE, "no float64 gradient: call pm_gradient_f64 or pm_set_mesh_f64 first");
    HIP_TRY(c, hipSetDevice(c->device));
    stage_timer t(c, PM_STAGE_GATHER);
    HIP_TRY(c, pm_launch_ngp_gather(c, 0.0, false, c->forces64));
    c->have_forces64 = true;
    return PM_OK;
}

int pm_get_forces_f64(pm_ctx *c, double *forces)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    if (!c->have_forces64) return fail(c, PM_ESTATE, "no float64 forces: call pm_forces_ngp_f64 first");
    if (!forces) return fail(c, PM_EINVAL, "forces is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n) {
        // storage order -> the caller's order, through the staging buffer (n * 24 bytes fit its n * 32)
        HIP_TRY(c, pm_launch_export_forces64(c, (double *)c->stage));
        HIP_TRY(c, hipMemcpyAsync(forces, c->stage, (size_t)c->n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return PM_OK;
}

// ---- the cloud-in-cell branch in float64 (PM_FLAG_F64) -------------------------------------------------------------

static int need_f64_particles(pm_ctx *c)
{
    TRY(whole_mesh(c));
    TRY(need_particles(c));
    if (!c->pos_f64 || !c->vel_f64) return fail(c, PM_ESTATE, "the float64 cloud-in-cell calls need float64 positions and velocities (pm_set_particles with PM_F64, PM_F64)");
    return PM_OK;
}

// CIC (functions.py:19-70) into a float64 mesh
int pm_deposit_f64(pm_ctx *c)
{
    TRY(need_f64_particles(c));
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(ensure_binned(c));
    TRY(dalloc(c, &c->density64, (size_t)c->g.M));
    stage_timer t(c, PM_STAGE_DEPOSIT);
    HIP_TRY(c, pm_launch_deposit64(c));
    return PM_OK;
}

// force_CIC (functions.py:99-180) on the float64 gradient, float64 forces (read with pm_get_forces_f64)
int pm_forces_f64(pm_ctx *c)
{
    TRY(need_f64_particles(c));
    if (!c->grad64[0]) return fail(c, PM_ESTATE, "no float64 gradient: call pm_gradient_f64 or pm_set_mesh_f64 first");
    HIP_TRY(c, hipSetDevice(c->device));
    stage_timer t(c, PM_STAGE_GATHER);
    HIP_TRY(c, pm_launch_gather64(c, 0.0, false, c->forces64));
    c->have_forces64 = true;
    return PM_OK;
}

// main.py:50-60 with every array in float64
static int step_f64(pm_ctx *c, double dt)
{
    TRY(pm_deposit_f64(c));
    TRY(pm_potential_f64(c));
    TRY(pm_gradient_f64(c));
    stage_timer t(c, PM_STAGE_GATHER);
    HIP_TRY(c, pm_launch_gather64(c, dt, true, nullptr));
    c->binned = false; c->have_forces = false; c->have_forces64 = false; c->keys_fresh = true;
    return PM_OK;
}

// main.py:50-60 with CIC_interpolation = False
static int step_ngp(pm_ctx *c, double dt)
{
    TRY(pm_deposit_ngp(c));
    TRY(pm_potential_f64(c));
    TRY(pm_gradient_f64(c));
    stage_timer t(c, PM_STAGE_GATHER);
    HIP_TRY(c, pm_launch_ngp_gather(c, dt, true, nullptr));
    c->pos_f64 = false;      // positions are float32 from the first drift on (main.py:60)
    c->binned = false; c->have_forces = false; c->have_forces64 = false; c->keys_fresh = false; c->order_known = false;
    return PM_OK;
}

int pm_set_flags(pm_ctx *c, unsigned flags)
{
    if (!c) return PM_EINVAL;
    if (flags & ~(unsigned)(PM_FLAG_TIMING | PM_FLAG_FUSED_GRADIENT | PM_FLAG_FULL_SORT))
        return fail(c, PM_EINVAL, "only PM_FLAG_TIMING, PM_FLAG_FUSED_GRADIENT and PM_FLAG_FULL_SORT can change");
    flags |= c->flags & (PM_FLAG_ROCFFT_ONLY | PM_FLAG_NGP | PM_FLAG_F64);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < pm_ctx::TIMED; ++s) pm_timer::flush(c, s);
    if (flags & PM_FLAG_TIMING) TRY(make_timing_events(c));
    c->flags = flags;
    return PM_OK;
}

int pm_stage_times(pm_ctx *c, float *ms, int *calls)
{
    if (!c) return PM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < PM_NSTAGES; ++s) {
        stage_timer::flush(c, s);
        if (ms) ms[s] = c->ms[s];
        if (calls) calls[s] = c->calls[s];
        c->ms[s] = 0.0f; c->calls[s] = 0;
    }
    return PM_OK;
}

int pm_kernel_times(pm_ctx *c, float *ms, int *calls)
{
    if (!c) return PM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < PM_NKERNELS; ++k) {
        const int s = PM_NSTAGES + k;
        pm_timer::flush(c, s);
        if (ms) ms[k] = c->ms[s];
        if (calls) calls[k] = c->calls[s];
        c->ms[s] = 0.0f; c->calls[s] = 0;
    }
    return PM_OK;
}

}  // extern "C"
