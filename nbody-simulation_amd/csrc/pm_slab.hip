// Slab decomposition: a context that owns planes [plane0, plane0 + nplanes) of the mesh and the
// particles whose cells lie in them.  Every call here does the local part of one stage of the step;
// what has to cross between slabs (the extra density plane, the transposes of the half spectrum
// around the x transform, the potential planes next to the slab, the particles that drift out)
// goes through device buffers the caller owns and exchanges.  Calls that fill or read such a buffer
// return after the work on the context's stream has finished.
#include "pm_internal.hpp"

#include <cmath>
#include <mutex>
#include <vector>

namespace {

int fail(pm_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

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

#define TRY(expr) do { int r_ = (expr); if (r_ != PM_OK) return r_; } while (0)

// A call that hands a device buffer back to the caller waits for the work that fills it -- unless the context
// works on the caller's own stream (pm_set_stream), which orders it with whatever the caller does next.
int done(pm_ctx *c)
{
    if (c->own_stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PM_OK;
}

// with PM_FLAG_TIMING: the launches of a call as one of the single kernels pm_kernel_times reports
struct kernel_timer : pm_timer { kernel_timer(pm_ctx *c, int k) : pm_timer(c, PM_NSTAGES + k) {} };

int need_slab(pm_ctx *c)
{
    if (!c) return PM_EINVAL;
    if (!c->slab) return fail(c, PM_ESTATE, "not a slab context: make it with pm_create_slab");
    return PM_OK;
}

// the words a drift starts from zero: outbox entries, overflow, entries per destination (one more word behind them is sticky)
size_t outbox_count_bytes(const pm_ctx *c) { return (size_t)(2 + c->nslabs) * sizeof(uint32_t); }

bool spectral_deposit(const pm_ctx *c) { return c->fft_custom && (c->flags & PM_FLAG_FUSED_GRADIENT); }

int bits_for(int64_t M)
{
    int b = 1;
    while (((int64_t)1 << b) < M) ++b;
    return b;
}

// rocFFT plans for grid sizes the hand-written kernels do not cover: batched 2-D real transforms of
// the owned planes, and the 1-D complex transform along x of the transposed spectrum (G, Gs, Ghp).
int make_plans(pm_ctx *c)
{
    const size_t G = (size_t)c->g.G, Gs = (size_t)c->slabn, P = (size_t)c->g.Ghp;
    rocfft_plan_description d = nullptr;
    const size_t len2[2] = {G, G};                                  // fastest axis first
    size_t rs[2] = {1, G}, cs[2] = {1, P}, off[1] = {0};
    FFT_TRY(c, rocfft_plan_description_create(&d));
    FFT_TRY(c, rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, off, off,
                                                       2, rs, G * G, 2, cs, G * P));
    FFT_TRY(c, rocfft_plan_create(&c->slab_fwd2, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_single,
                                  2, len2, Gs, d));
    rocfft_plan_description_destroy(d);
    FFT_TRY(c, rocfft_plan_description_create(&d));
    FFT_TRY(c, rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, off, off,
                                                       2, cs, G * P, 2, rs, G * G));
    FFT_TRY(c, rocfft_plan_create(&c->slab_inv2, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, rocfft_precision_single,
                                  2, len2, Gs, d));
    rocfft_plan_description_destroy(d);
    const size_t len1[1] = {G};
    size_t xs[1] = {Gs * P};
    for (int inv = 0; inv < 2; ++inv) {
        FFT_TRY(c, rocfft_plan_description_create(&d));
        FFT_TRY(c, rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, off, off,
                                                           1, xs, 1, 1, xs, 1));
        FFT_TRY(c, rocfft_plan_create(inv ? &c->slab_invx : &c->slab_fwdx, rocfft_placement_inplace,
                                      inv ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward,
                                      rocfft_precision_single, 1, len1, Gs * P, d));
        rocfft_plan_description_destroy(d);
    }
    size_t need = 0;
    for (rocfft_plan p : {c->slab_fwd2, c->slab_inv2, c->slab_fwdx, c->slab_invx}) {
        size_t w = 0;
        FFT_TRY(c, rocfft_plan_get_work_buffer_size(p, &w));
        need = w > need ? w : need;
    }
    if (need > c->fft_work_bytes) {
        if (c->fft_work) hipFree(c->fft_work);
        c->fft_work = nullptr;
        HIP_TRY(c, hipMalloc(&c->fft_work, need));
        c->fft_work_bytes = need;
        FFT_TRY(c, rocfft_execution_info_set_work_buffer(c->info, c->fft_work, c->fft_work_bytes));
    }
    return PM_OK;
}

int ensure_binned(pm_ctx *c)
{
    if (c->binned) return PM_OK;
    HIP_TRY(c, pm_settle_velocities(c));
    if (!c->keys_fresh) HIP_TRY(c, pm_launch_cell_keys(c));
    // one key more than the slab has cells: "not mine" -- which a slab that is the whole mesh never hands out
    TRY(pm_bin_order(c, bits_for((int64_t)c->g.Gs * c->g.G * c->g.G + (c->g.Gs != c->g.G ? 1 : 0))));
    HIP_TRY(c, pm_launch_reorder(c, !c->last_incremental));     // (with the row starts, unless the binning from the order before has brought them up to date itself)
    c->n -= c->leaving;         // they carry the last key: the tail of the cell order
    c->leaving = 0;
    c->binned = true;
    c->order_known = true;
    c->order_n = c->n;
    c->have_forces = false; c->have_forces64 = false;      // indexed by the storage slots of before the reorder
    return PM_OK;
}

}  // namespace

extern "C" {

pm_ctx *pm_create_slab_windowed(int grid_size, double box_size, int64_t n_capacity, int device, unsigned flags, int plane0, int nplanes,
                                int window_planes)
{
    if (nplanes < 1 || plane0 < 0 || grid_size % nplanes != 0 || plane0 % nplanes != 0 || plane0 + nplanes > grid_size ||
        (nplanes != grid_size && nplanes + 3 > grid_size)) {
        pm_set_create_error("pm_create_slab: nplanes must divide grid_size, plane0 must be a multiple of nplanes, and nplanes + 3 <= grid_size");
        return nullptr;
    }
    // the hand-written transforms cover whole groups of 16 rows
    if (pm_fft_custom_supported(grid_size) && ((int64_t)nplanes * grid_size) % 16 != 0) flags |= PM_FLAG_ROCFFT_ONLY;
    // PM_FLAG_FUSED_GRADIENT on a slab: the deposit leaves z-transformed rows (hand-written transforms only); the
    // gradient is always taken inside the gather
    if (!pm_fft_custom_supported(grid_size) || (flags & PM_FLAG_ROCFFT_ONLY)) flags &= ~(unsigned)PM_FLAG_FUSED_GRADIENT;
    if (window_planes != nplanes) {
        // a window's planes travel to the slabs that transform them as z-transformed planes: the deposit must write those
        if (!(flags & PM_FLAG_FUSED_GRADIENT)) {
            pm_set_create_error("pm_create_slab_windowed: a particle window other than the slab needs PM_FLAG_FUSED_GRADIENT on a grid size of the hand-written transforms");
            return nullptr;
        }
        if (window_planes < nplanes || window_planes + 3 > grid_size) {
            pm_set_create_error("pm_create_slab_windowed: window_planes must be in [nplanes, grid_size - 3]");
            return nullptr;
        }
    }
    pm_ctx *c = pm_create_planes(grid_size, box_size, n_capacity, device, flags, plane0, nplanes, true, window_planes);
    if (!c) return nullptr;
    c->y_in_exchange = c->fft_custom && (nplanes & (nplanes - 1)) == 0;
    auto setup = [&]() -> int {
        HIP_TRY(c, hipSetDevice(device));
        // (when the windows are cut anew, most of a context's particles may leave at once)
        c->outbox_cap = window_planes != nplanes ? c->cap : c->cap / 4 + 1024;
        HIP_TRY(c, hipMalloc((void **)&c->outbox, (size_t)c->outbox_cap * sizeof(pm_migrant)));
        HIP_TRY(c, hipMalloc((void **)&c->outbox_key, (size_t)c->outbox_cap * sizeof(uint64_t)));
        HIP_TRY(c, hipMalloc((void **)&c->outbox_key2, (size_t)c->outbox_cap * sizeof(uint64_t)));
        HIP_TRY(c, hipMalloc((void **)&c->outbox_idx2, (size_t)c->outbox_cap * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void **)&c->outbox_offsets, (size_t)(c->nslabs + 1) * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void **)&c->outbox_count, outbox_count_bytes(c) + sizeof(uint32_t)));
        HIP_TRY(c, hipMemset(c->outbox_count, 0, outbox_count_bytes(c) + sizeof(uint32_t)));
        c->outbox_tmp_bytes = pm_sort_outbox_tmp_bytes(c->outbox_cap);
        HIP_TRY(c, hipMalloc(&c->outbox_tmp, c->outbox_tmp_bytes + 16));
        // the windows to begin with: every rank's own slab
        std::vector<int> cuts((size_t)c->nslabs + 1);
        for (int r = 0; r <= c->nslabs; ++r) cuts[(size_t)r] = r * nplanes;
        HIP_TRY(c, hipMalloc((void **)&c->cuts, cuts.size() * sizeof(int)));
        HIP_TRY(c, hipMemcpy(c->cuts, cuts.data(), cuts.size() * sizeof(int), hipMemcpyHostToDevice));
        if (!c->fft_custom) TRY(make_plans(c));
        return PM_OK;
    };
    if (setup() != PM_OK) {
        pm_set_create_error(c->err);
        pm_destroy(c);
        return nullptr;
    }
    return c;
}

pm_ctx *pm_create_slab(int grid_size, double box_size, int64_t n_capacity, int device, unsigned flags, int plane0, int nplanes)
{
    return pm_create_slab_windowed(grid_size, box_size, n_capacity, device, flags, plane0, nplanes, nplanes);
}

// The planes whose particles this context holds from now on: [cuts[rank], cuts[rank + 1]) of cuts[0] = 0 < cuts[1] < ... <
// cuts[ranks] = G, the windows of all ranks (a particle that leaves the window goes to the rank whose window holds its
// new cell).  Particles already in the context stay where they are until pm_slab_rekey sorts out which of them still
// belong here.
int pm_slab_set_window(pm_ctx *c, const int *cuts, int ranks, int rank)
{
    TRY(need_slab(c));
    if (!cuts || ranks != c->nslabs || rank != c->slab0 / c->slabn) return fail(c, PM_EINVAL, "pm_slab_set_window: cuts of all G / nplanes ranks, and this context's own rank");
    if (cuts[0] != 0 || cuts[ranks] != c->g.G) return fail(c, PM_EINVAL, "pm_slab_set_window: cuts must run from 0 to grid_size");
    for (int r = 0; r < ranks; ++r)
        if (cuts[r + 1] <= cuts[r]) return fail(c, PM_EINVAL, "pm_slab_set_window: every rank needs at least one plane");
    const int w0 = cuts[rank], wn = cuts[rank + 1] - cuts[rank];
    if (wn > c->wmax) return fail(c, PM_EINVAL, "pm_slab_set_window: " + std::to_string(wn) + " planes, the context has room for " + std::to_string(c->wmax));
    if (c->wmax == c->slabn && (w0 != c->slab0 || wn != c->slabn)) return fail(c, PM_ESTATE, "this context's window is its slab: make it with pm_create_slab_windowed");
    if (c->outbox_pending) return fail(c, PM_ESTATE, "pm_slab_set_window belongs behind the emptying of the outbox");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->cuts, cuts, (size_t)(ranks + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, pm_settle_velocities(c));       // (src belongs to the order of the old window)
    c->g.i0 = w0; c->g.Gs = wn; c->g.Gd = wn + 1;
    // cell keys count from the window's first plane: nothing computed for the old window holds
    c->binned = false; c->keys_fresh = false; c->order_known = false; c->movers_marked = false; c->have_forces = false;
    return PM_OK;
}

int pm_slab_window(const pm_ctx *c, int *plane0, int *nplanes, int *max_planes)
{
    if (!c || !c->slab) return PM_EINVAL;
    if (plane0) *plane0 = c->g.i0;
    if (nplanes) *nplanes = c->g.Gs;
    if (max_planes) *max_planes = c->wmax;
    return PM_OK;
}

// After pm_slab_set_window: the particles whose cell lies outside the new window go to the outbox, as the particles
// that leave in a drift do (pm_slab_outbox_counts / pm_slab_outbox_records / pm_slab_inbox follow).
int pm_slab_rekey(pm_ctx *c)
{
    TRY(need_slab(c));
    if (!c->pf[0]) return fail(c, PM_ESTATE, "no particles");
    if (c->outbox_pending) return fail(c, PM_ESTATE, "the outbox of the last drift has not been taken yet");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->leaving) TRY(ensure_binned(c));     // (drops the leavers of a drift that are still among the slots)
    HIP_TRY(c, pm_settle_velocities(c));
    HIP_TRY(c, hipMemsetAsync(c->outbox_count, 0, outbox_count_bytes(c), c->stream));
    HIP_TRY(c, pm_launch_rekey(c));
    c->binned = false; c->have_forces = false; c->keys_fresh = true; c->order_known = false; c->movers_marked = false;
    c->outbox_pending = true;
    return PM_OK;
}

// Live particles per mesh plane, ADDED to the grid_size counters at device_counts (uint64): what the windows are cut by.
int pm_slab_plane_counts(pm_ctx *c, uint64_t *device_counts)
{
    if (!c || !device_counts) return PM_EINVAL;
    if (!c->pf[0]) return fail(c, PM_ESTATE, "no particles");
    if (c->outbox_pending) return fail(c, PM_ESTATE, "the outbox of the last drift has not been taken yet");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->leaving) TRY(ensure_binned(c));
    HIP_TRY(c, pm_launch_plane_counts(c, (unsigned long long *)device_counts));
    TRY(done(c));
    return PM_OK;
}

// pos, vel: host (n,3) float32; mass: host (n) or NULL; ids: host (n) particle numbers of the whole run
// (they travel with the particles) or NULL for 0 .. n-1.  Every particle's cell must lie in the slab.
int pm_slab_set_particles(pm_ctx *c, const float *pos, const float *vel, const float *mass, const uint32_t *ids, int64_t n)
{
    TRY(need_slab(c));
    TRY(pm_set_particles(c, pos, PM_F32, vel, PM_F32, mass, n));
    if (ids && n > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->stage, ids, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, pm_launch_set_ids(c, (const uint32_t *)c->stage));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->leaving = 0;
    c->outbox_pending = false;
    c->slab_check = true;
    HIP_TRY(c, hipMemsetAsync(c->outbox_count, 0, outbox_count_bytes(c), c->stream));
    return PM_OK;
}

// Cloud-in-cell deposit of the owned particles into planes 0 .. nplanes of the context's density
// (nplanes + 1 planes: the last one is plane 0 of the next slab).
int pm_slab_deposit(pm_ctx *c)
{
    TRY(need_slab(c));
    if (c->outbox_pending) return fail(c, PM_ESTATE, "pm_slab_outbox (or pm_slab_outbox_records) must follow pm_slab_gather_kick_drift before the next deposit");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(ensure_binned(c));
    if (c->slab_check) {
        uint32_t mine = 0;
        HIP_TRY(c, hipMemcpyAsync(&mine, c->row_start + (size_t)c->g.Gs * c->g.G, sizeof(mine), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if ((int64_t)mine != c->n) return fail(c, PM_EINVAL, std::to_string(c->n - (int64_t)mine) + " particles given to this slab lie in other slabs");
        c->slab_check = false;
    }
    HIP_TRY(c, pm_launch_deposit(c, spectral_deposit(c)));
    return PM_OK;
}

// Planes [first, first + count) of a mesh of the context <-> a device buffer of count*G*G floats.
// which: PM_MESH_DENSITY (planes 0 .. nplanes) or PM_MESH_POTENTIAL (plane 0 is mesh plane plane0 - 1,
// planes 1 .. nplanes the owned ones, then two more).  op 0: mesh -> buffer, 1: buffer -> mesh, 2: mesh += buffer.
int pm_slab_planes(pm_ctx *c, int which, int first, int count, void *device_buffer, int op)
{
    TRY(need_slab(c));
    const bool zs = which == PM_SLAB_ZSPECTRUM || which == PM_SLAB_MESH_ZSPECTRUM;
    const int limit = which == PM_MESH_POTENTIAL ? c->g.Gs + 3 : (which == PM_SLAB_MESH_ZSPECTRUM || which == PM_SLAB_MESH_POTENTIAL) ? c->slabn : c->g.Gs + 1;
    if ((which != PM_MESH_DENSITY && which != PM_MESH_POTENTIAL && !zs && which != PM_SLAB_MESH_POTENTIAL) || first < 0 || count < 0 ||
        first + count > limit || !device_buffer || op < 0 || op > 2)
        return fail(c, PM_EINVAL, "pm_slab_planes: bad mesh, plane range, buffer or op");
    if (zs && !spectral_deposit(c))
        return fail(c, PM_ESTATE, "pm_slab_planes: this context's deposit writes density planes (PM_MESH_DENSITY), not z-transformed ones");
    if (which == PM_MESH_DENSITY && spectral_deposit(c))
        return fail(c, PM_ESTATE, "pm_slab_planes: this context's deposit writes z-transformed planes (PM_SLAB_ZSPECTRUM), no density mesh");
    HIP_TRY(c, hipSetDevice(c->device));
    // a plane of the half spectrum is G rows of pitch complex64 = 2 G pitch floats (the padding of the rows travels along)
    const size_t plane = zs ? 2 * (size_t)c->g.G * c->g.Ghp : (size_t)c->g.G * c->g.G;
    float *base = which == PM_MESH_DENSITY ? c->density : which == PM_MESH_POTENTIAL ? c->potential : which == PM_SLAB_ZSPECTRUM ? (float *)c->spectrum
                : which == PM_SLAB_MESH_ZSPECTRUM ? (float *)c->mesh_spec : c->mesh_pot;
    float *mesh = base + (size_t)first * plane;
    const size_t bytes = (size_t)count * plane * sizeof(float);
    if (op == 0) HIP_TRY(c, hipMemcpyAsync(device_buffer, mesh, bytes, hipMemcpyDeviceToDevice, c->stream));
    else if (op == 1) HIP_TRY(c, hipMemcpyAsync(mesh, device_buffer, bytes, hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(c, pm_launch_add(c, mesh, (const float *)device_buffer, (int64_t)count * (int64_t)plane));
    TRY(done(c));
    return PM_OK;
}

// Planes between the window's buffers and the slab's own, inside the context: to_mesh != 0 copies (add == 0) or adds
// z-transformed planes [first, first + count) of the window's deposit to the slab's planes from `mesh_first` on; to_mesh
// == 0 copies potential planes [mesh_first, ...) of the slab to planes `first` ... of the window's potential buffer.  For
// the part of a window that lies in the context's own slab, which needs no exchange.
int pm_slab_copy_planes(pm_ctx *c, int to_mesh, int first, int mesh_first, int count, int add)
{
    TRY(need_slab(c));
    if (count == 0) return PM_OK;
    if (c->wmax == c->slabn) return fail(c, PM_ESTATE, "pm_slab_copy_planes: this context's window is its slab (the buffers are the same memory)");
    const int limit = to_mesh ? c->g.Gs + 1 : c->g.Gs + 3;
    if (first < 0 || count < 0 || first + count > limit || mesh_first < 0 || mesh_first + count > c->slabn)
        return fail(c, PM_EINVAL, "pm_slab_copy_planes: plane range outside the buffers");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t plane = (size_t)c->g.G * c->g.G, splane = 2 * (size_t)c->g.G * c->g.Ghp;
    if (to_mesh) {
        float *dst = (float *)c->mesh_spec + (size_t)mesh_first * splane;
        const float *src = (const float *)c->spectrum + (size_t)first * splane;
        if (add) HIP_TRY(c, pm_launch_add(c, dst, src, (int64_t)count * (int64_t)splane));
        else HIP_TRY(c, hipMemcpyAsync(dst, src, (size_t)count * splane * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else {
        HIP_TRY(c, hipMemcpyAsync(c->potential + (size_t)first * plane, c->mesh_pot + (size_t)mesh_first * plane, (size_t)count * plane * sizeof(float),
                                  hipMemcpyDeviceToDevice, c->stream));
    }
    TRY(done(c));
    return PM_OK;
}

// dir < 0: density planes 0 .. nplanes-1 -> half spectrum (nplanes, G, pitch), transformed along z and y.
// dir > 0: half spectrum -> potential planes 1 .. nplanes (unnormalised inverse along y and z).
int pm_slab_fft2(pm_ctx *c, int dir)
{
    TRY(need_slab(c));
    HIP_TRY(c, hipSetDevice(c->device));
    // (the slab's own planes: where the window is the slab, mesh_spec is spectrum and mesh_pot plane 1 of potential)
    float *pot = c->mesh_pot;
    if (c->fft_custom) {
        // (with y_in_exchange the passes along y belong to pm_slab_pack / pm_slab_unpack)
        if (dir < 0) {
            // (a deposit that writes z-transformed rows has done the transform along z already)
            if (!spectral_deposit(c)) HIP_TRY(c, pm_launch_row_fft_planes(c, true, c->density, nullptr, c->mesh_spec, c->slabn));
            if (!c->y_in_exchange) HIP_TRY(c, pm_launch_col_fft_planes(c, -1, c->mesh_spec, c->slabn));
        } else {
            if (!c->y_in_exchange) HIP_TRY(c, pm_launch_col_fft_planes(c, 1, c->mesh_spec, c->slabn));
            kernel_timer k(c, PM_KERNEL_ROW_INV);
            HIP_TRY(c, pm_launch_row_fft_planes(c, false, nullptr, pot, c->mesh_spec, c->slabn));
        }
    } else {
        void *in[1], *out[1];
        if (dir < 0) { in[0] = c->density; out[0] = c->mesh_spec; FFT_TRY(c, rocfft_execute(c->slab_fwd2, in, out, c->info)); }
        else { in[0] = c->mesh_spec; out[0] = pot; FFT_TRY(c, rocfft_execute(c->slab_inv2, in, out, c->info)); }
    }
    TRY(done(c));
    return PM_OK;
}

// Half spectrum of the slab -> nslabs blocks for the all-to-all (block p: rows p*nplanes .. of every
// owned plane), and back.  The buffers hold nplanes * G * pitch complex64.
int pm_slab_pack(pm_ctx *c, void *device_blocks)
{
    TRY(need_slab(c));
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->y_in_exchange) {
        kernel_timer k(c, PM_KERNEL_COL_Y_FWD);
        HIP_TRY(c, pm_launch_col_fft_blocks(c, -1, c->mesh_spec, (float2 *)device_blocks, c->slabn, 0, c->slabn));
    } else HIP_TRY(c, pm_launch_transpose_pack(c, c->mesh_spec, (float2 *)device_blocks, false));
    TRY(done(c));
    return PM_OK;
}

int pm_slab_unpack(pm_ctx *c, const void *device_blocks)
{
    TRY(need_slab(c));
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->y_in_exchange) {
        kernel_timer k(c, PM_KERNEL_COL_Y_INV);
        HIP_TRY(c, pm_launch_col_fft_blocks(c, 1, c->mesh_spec, (float2 *)device_blocks, c->slabn, 0, c->slabn));
    } else HIP_TRY(c, pm_launch_transpose_pack(c, (const float2 *)device_blocks, c->mesh_spec, true));
    TRY(done(c));
    return PM_OK;
}

// The same for owned planes [first, first + count) only, so that the all-to-all of some planes can run beside the
// transforms of others: forward = transforms along z (unless the deposit did them) and y into those planes' part of the
// blocks; inverse = from their part of the blocks back along y and z into potential planes 1 + first ....
static int planes_in_turn(pm_ctx *c, int first, int count)
{
    TRY(need_slab(c));
    if (!(c->fft_custom && c->y_in_exchange))
        return fail(c, PM_ESTATE, "plane-by-plane transforms need the hand-written transforms and a power-of-two number of planes: use pm_slab_fft2 with pm_slab_pack / pm_slab_unpack");
    if (first < 0 || count < 0 || first + count > c->slabn) return fail(c, PM_EINVAL, "plane range outside the slab");
    return PM_OK;
}

int pm_slab_forward_planes(pm_ctx *c, void *device_blocks, int first, int count)
{
    TRY(planes_in_turn(c, first, count));
    if (!device_blocks) return fail(c, PM_EINVAL, "NULL buffer");
    if (count == 0) return PM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t plane = (size_t)c->g.G * c->g.G, splane = (size_t)c->g.G * c->g.Ghp;
    if (!spectral_deposit(c)) HIP_TRY(c, pm_launch_row_fft_planes(c, true, c->density + first * plane, nullptr, c->mesh_spec + first * splane, count));
    {
        kernel_timer k(c, PM_KERNEL_COL_Y_FWD);
        HIP_TRY(c, pm_launch_col_fft_blocks(c, -1, c->mesh_spec, (float2 *)device_blocks, c->slabn, first, count));
    }
    TRY(done(c));
    return PM_OK;
}

int pm_slab_inverse_planes(pm_ctx *c, const void *device_blocks, int first, int count)
{
    TRY(planes_in_turn(c, first, count));
    if (!device_blocks) return fail(c, PM_EINVAL, "NULL buffer");
    if (count == 0) return PM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t plane = (size_t)c->g.G * c->g.G, splane = (size_t)c->g.G * c->g.Ghp;
    {
        kernel_timer k(c, PM_KERNEL_COL_Y_INV);
        HIP_TRY(c, pm_launch_col_fft_blocks(c, 1, c->mesh_spec, (float2 *)device_blocks, c->slabn, first, count));
    }
    {
        kernel_timer k(c, PM_KERNEL_ROW_INV);
        HIP_TRY(c, pm_launch_row_fft_planes(c, false, nullptr, c->mesh_pot + first * plane, c->mesh_spec + first * splane, count));
    }
    TRY(done(c));
    return PM_OK;
}

// device_spectrum_t: (G, nplanes, pitch) complex64 -- every plane's rows plane0 .. plane0 + nplanes - 1, as the
// all-to-all leaves them.  Forward transform along x, 1/k^2 and 1/M (functions.py:75-89), inverse along x, in place.
int pm_slab_solve_x(pm_ctx *c, void *device_spectrum_t)
{
    TRY(need_slab(c));
    if (!device_spectrum_t) return fail(c, PM_EINVAL, "NULL buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    float2 *s = (float2 *)device_spectrum_t;
    if (c->fft_custom) {
        kernel_timer k(c, PM_KERNEL_COL_X_GREEN);
        HIP_TRY(c, pm_launch_solve_x(c, s, c->slab0, c->slabn));
    } else {
        void *io[1] = {s};
        FFT_TRY(c, rocfft_execute(c->slab_fwdx, io, nullptr, c->info));
        HIP_TRY(c, pm_launch_green_t(c, s, c->slab0, c->slabn));
        FFT_TRY(c, rocfft_execute(c->slab_invx, io, nullptr, c->info));
    }
    TRY(done(c));
    return PM_OK;
}

// force_CIC with np.gradient evaluated on the fly from the potential planes (all nplanes + 3 must be in
// place), kick, drift, wrap (main.py:54-60).  Particles whose new cell lies in another slab go to the outbox.
int pm_slab_gather_kick_drift(pm_ctx *c, double dt)
{
    TRY(need_slab(c));
    if (!c->pf[0]) return fail(c, PM_ESTATE, "no particles");
    if (c->pos_f64 || c->vel_f64) return fail(c, PM_ESTATE, "slab contexts hold float32 particles");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->outbox_count, 0, outbox_count_bytes(c), c->stream));
    {
        kernel_timer k(c, PM_KERNEL_GATHER);
        HIP_TRY(c, pm_launch_gather(c, dt, true, false, true));
    }
    c->binned = false; c->have_forces = false; c->keys_fresh = true;
    c->outbox_pending = true;       // the leavers are still in the live set until the outbox has been taken
    return PM_OK;
}

// How many particles left in the last drift for each slab, as nslabs + 1 int64 in DEVICE memory: counts[d] for
// destination d, then the number that did not fit the outbox (must be 0).  Queued on the context's stream.
int pm_slab_outbox_counts(pm_ctx *c, int64_t *device_counts)
{
    TRY(need_slab(c));
    if (!device_counts) return fail(c, PM_EINVAL, "NULL device_counts");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, pm_launch_outbox_counts(c, device_counts));
    TRY(done(c));
    return PM_OK;
}

// The leavers as 32-byte records {x, y, z, m, vx, vy, vz, id} sorted by (destination slab, id) into device_records,
// given the counts pm_slab_outbox_counts reported (host, nslabs entries): nothing is read back from the device here.
int pm_slab_outbox_records(pm_ctx *c, const int64_t *counts, void *device_records, int64_t capacity)
{
    TRY(need_slab(c));
    if (!counts) return fail(c, PM_EINVAL, "NULL counts");
    int64_t total = 0;
    for (int d = 0; d < c->nslabs; ++d) {
        if (counts[d] < 0) return fail(c, PM_EINVAL, "negative count");
        total += counts[d];
    }
    if (total > c->outbox_cap) return fail(c, PM_EINVAL, "more records than the outbox holds");
    if (total > capacity) return fail(c, PM_EINVAL, "device_records too small for " + std::to_string(total) + " records");
    HIP_TRY(c, hipSetDevice(c->device));
    c->leaving = total;
    c->outbox_pending = false;
    if (total) {
        if (!device_records) return fail(c, PM_EINVAL, "NULL device_records");
        HIP_TRY(c, pm_sort_outbox(c, (uint32_t)total));
        HIP_TRY(c, pm_launch_pack_outbox(c, (uint32_t)total, (pm_migrant *)device_records));
    }
    TRY(done(c));
    return PM_OK;
}

// The two calls above in one, with the counts read back here (the host waits for the drift): counts[d] = how many go
// to slab d (host, nslabs entries).
int pm_slab_outbox(pm_ctx *c, int64_t *counts, void *device_records, int64_t capacity)
{
    TRY(need_slab(c));
    if (!counts) return fail(c, PM_EINVAL, "NULL counts");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<uint32_t> h((size_t)c->nslabs + 2, 0u);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->outbox_count, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<uint32_t> sticky(1, 0u);
    HIP_TRY(c, hipMemcpy(sticky.data(), c->outbox_count + 2 + c->nslabs, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (sticky[0]) return fail(c, PM_ESTATE, "a particle was found in this slab whose cell belongs to another (a record delivered to the wrong slab)");
    if (h[1]) return fail(c, PM_ENOMEM, "outbox overflow: " + std::to_string(h[0]) + " particles left the slab in one step");
    for (int d = 0; d < c->nslabs; ++d) counts[d] = (int64_t)h[2 + d];
    return pm_slab_outbox_records(c, counts, device_records, capacity);
}

// Particles arriving from other slabs (records as pm_slab_outbox writes them) join the live set.
int pm_slab_inbox(pm_ctx *c, const void *device_records, int64_t n)
{
    TRY(need_slab(c));
    if (n < 0 || (n > 0 && !device_records)) return fail(c, PM_EINVAL, "bad inbox");
    if (c->outbox_pending) return fail(c, PM_ESTATE, "pm_slab_inbox belongs behind pm_slab_outbox (or pm_slab_outbox_records)");
    HIP_TRY(c, hipSetDevice(c->device));
    // the leavers keep their slots until the next binning: when the arrivals do not fit beside them (a change of windows
    // moves most of a context's particles), the binning that drops them runs now
    if (c->n + n > c->cap && c->leaving > 0) TRY(ensure_binned(c));
    if (c->n + n > c->cap) return fail(c, PM_ENOMEM, "particle capacity exceeded: " + std::to_string(c->n + n) + " > " + std::to_string(c->cap));
    if (c->v_stale) HIP_TRY(c, pm_settle_velocities(c));      // (after that binning the velocity records lag behind: the arrivals' go in behind the settled ones)
    HIP_TRY(c, pm_launch_append(c, (const pm_migrant *)device_records, n));
    TRY(done(c));
    c->n += n;
    c->binned = false;
    return PM_OK;
}

// Owned particles in storage order: ids (n) uint32, pos and vel (n,3) float32, host buffers of at least
// pm_num_particles() entries as it was before the call; afterwards pm_num_particles() says how many were
// written (particles that left in the last drift are dropped here at the latest).
int pm_slab_get_particles(pm_ctx *c, uint32_t *ids, float *pos, float *vel)
{
    TRY(need_slab(c));
    if (!ids || !pos || !vel) return fail(c, PM_EINVAL, "NULL buffer");
    if (c->outbox_pending) return fail(c, PM_ESTATE, "the last drift's leavers are still among the particles: take the outbox first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->leaving) TRY(ensure_binned(c));
    const size_t n = (size_t)c->n;
    if (!n) return PM_OK;
    uint32_t *di = (uint32_t *)c->stage;
    float *dp = (float *)(di + n), *dv = dp + 3 * n;       // 28 n bytes of the 32 n staging bytes
    HIP_TRY(c, pm_launch_export_slab(c, di, dp, dv));
    HIP_TRY(c, hipMemcpyAsync(ids, di, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(pos, dp, 3 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(vel, dv, 3 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PM_OK;
}

// Row pitch (complex64 elements) of the half spectrum buffers of this context.
int pm_slab_pitch(const pm_ctx *c) { return c ? c->g.Ghp : -1; }

}  // extern "C"
