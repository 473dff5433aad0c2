// Internal declarations shared by the translation units of libpm.so.
#pragma once

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "pm.h"

// Geometry handed to every kernel by value.
struct pm_geom {
    int G;          // cells per side
    int Gh;         // G/2 + 1: length of the half-spectrum's fastest axis
    int Ghp;        // row pitch of the half spectrum in complex elements (>= Gh)
    int64_t M;      // G^3
    double L;       // box size
    double inv;     // G / L, the float64 quotient of functions.py:23
    float Lf;       // (float)L for the float32 remainder of main.py:60
    // Slab of the mesh this context owns (the whole mesh unless it was made by pm_create_slab):
    int i0;         // first owned plane
    int Gs;         // owned planes; a particle is owned by the slab that holds its cell's plane i
    int Gd;         // planes the deposit writes: G (periodic along i), or Gs + 1 (the last one belongs to the next slab)
    int lo;         // planes of the potential buffer below plane i0 (0, or 1 in a slab)
    int sb;         // half spectrum: rows of 2^sb neighbouring planes are interleaved (pm_spec_row); 0 = plain (i, j) order
    float soft;     // Gaussian softening: the Green's function is 1/k2 * exp(-soft * k2), k2 in (cycles/sample)^2; 0 = the bare 1/k2 of functions.py:85
};

// Row (i, j) of the half spectrum is row pm_spec_row(...) of the buffer (times the pitch).  With sb = 0 that is
// i * G + j.  Whole-mesh contexts on the hand-written transforms use sb > 0: the rows j of planes i0 .. i0 + 2^sb - 1
// alternate, so that a transform along i steps through memory in strides of one row for 2^sb rows at a time instead
// of G rows every time (a 16-column tile of the pass along i then touches G / 2^sb regions of memory, not G: its
// address translations fit the TLBs), while the pass along j still spans only 2^sb planes' worth of memory.
__host__ __device__ inline int64_t pm_spec_row(int G, int sb, int i, int j)
{
    return ((((int64_t)(i >> sb) * G + j) << sb) | (int64_t)(i & ((1 << sb) - 1)));
}

// What leaves a slab: a particle's position and velocity records.
struct pm_migrant { float x, y, z, m, vx, vy, vz; uint32_t id; };

// Particle records.  A particle is two 16-byte records (32-byte ones while a coordinate is
// float64): position + mass, and velocity + the particle's index in the caller's order.  Every
// kernel moves a record with one or two 16-byte accesses, and the reorder into cell order
// touches two lines per particle instead of eight.
template <typename T> struct pm_prec;
template <> struct alignas(16) pm_prec<float> { float x, y, z, m; };
template <> struct alignas(16) pm_prec<double> { double x, y, z; float m; float pad; };
// What the deposit reads of a particle besides its cell key: offsets inside the cell and the mass.
struct alignas(16) pm_frac { float dx, dy, dz, m; };
template <typename T> struct pm_vrec;
template <> struct alignas(16) pm_vrec<float> { float x, y, z; uint32_t id; };
template <> struct alignas(16) pm_vrec<double> { double x, y, z; uint32_t id; uint32_t pad; };

// words of pm_ctx::rb_ctl (pm_rebin.hip): the ticket that hands out tiles, and the movers found (the device-side count
// of the sort that follows)
enum { PM_RB_TICKET = 0, PM_RB_MOVERS = 1, PM_RB_WORDS = 8 };

constexpr int64_t PM_REBIN_MIN = 1 << 20;       // particle sets below this are always sorted whole (pm_ctx::rebin_min: the environment may say otherwise)
// What the drift reports about the particles that changed their cell.  mask (may be null): a bit per storage slot, in
// words of 64 slots, set where the particle's new cell key differs from its old one.  A sample of the workgroups also
// counts them for the host: *count holds the waves that have reported (upper half) and the movers they saw (lower
// half); the last of `expect` waves writes the estimate for the whole set to *seen (pinned host memory) and clears the
// word.  expect == 0: nothing is counted.
struct pm_mover_probe { unsigned long long *count; volatile uint32_t *seen; uint32_t expect; uint64_t *mask; uint32_t every; };

struct pm_ctx {
    pm_geom g{};
    int device = 0;
    // pm_run: snapshot buffers (positions in the caller's order), the stream their copies to the host run on
    float *snap[2] = {nullptr, nullptr};
    hipStream_t copy = nullptr;
    hipEvent_t snap_ready[2] = {nullptr, nullptr}, snap_done[2] = {nullptr, nullptr};
    unsigned flags = 0;
    int64_t cap = 0;    // particle capacity
    int64_t n = 0;      // particles held
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;         // a second stream: the units of crowded rows are summed beside the ordinary rows
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Four pinned host words that take turns: deposit number d (counted over the life of the context) writes d + 1 and how many
    // units it planned, or would have planned, to word d % 4 -- one 8-byte store by its last kernel -- and deposit d + 2 reads
    // that word, waiting for the number if it is not there yet (it is: that deposit finished a step ago), and plans units or
    // not by the count.  The same deposit of the same run therefore always decides the same way, whatever the host's lead
    // over the device: which of the two ways a deposit sums a crowded row shows in the last bits.
    volatile unsigned long long *units_seen = nullptr;
    uint64_t deposits = 0;                      // deposits of this context so far
    uint64_t deposits_of_set = 0;               // ... since the particle set was given (its first two plan units)
    bool own_stream = true;     // false after pm_set_stream: the caller's stream, which orders this context's work with the caller's

    // Particle state, double-buffered: `cur` is the live set, 1-cur the target of the next
    // reorder.  Positions are float32 except between pm_set_particles(PM_F64) and the first
    // drift, when pd[] holds them.
    int cur = 0;
    int vcur = 0;               // live velocity buffer
    bool v_stale = false;       // velocity slot r still sits at slot src[r] of vf/vd[vcur] (one reorder behind)
    pm_prec<float> *pf[2] = {};
    pm_prec<double> *pd[2] = {};
    pm_vrec<float> *vf[2] = {};
    pm_vrec<double> *vd[2] = {};
    pm_frac *frac = nullptr;    // fraction records of the live set, written by the reorder pass
    bool pos_f64 = false;       // live positions are in pd[]
    bool pos_in_f64 = false;    // dtype given to pm_set_particles
    bool vel_f64 = false;
    bool binned = false;        // live set is in cell order and row_start matches it
    bool have_forces = false;
    bool have_forces64 = false;         // forces64 holds force_NGP of the float64 gradient
    bool keys_fresh = false;    // key_in holds the cell keys of the live positions (written by the last drift)
    unsigned mesh_valid = 0;    // bit PM_MESH_*: that float32 mesh holds what a stage or pm_set_mesh wrote (a fused step writes neither density nor gradients)
    double *elog = nullptr;     // pm_run: (samples, 2) energies on the device, kept between calls
    size_t elog_cap = 0;        // samples it holds

    // binning scratch
    uint32_t *key_in = nullptr, *key_out = nullptr, *src = nullptr;
    // binning from the cell order of the step before (pm_rebin.hip; buffers allocated on first use)
    int64_t rebin_min = PM_REBIN_MIN;   // PM_REBIN_MIN, or what the environment variable of that name said when the context was made
    bool last_incremental = false;      // the last binning started from the order before it
    bool movers_marked = false;         // rb_mask holds the movers of the last drift (for the keys in key_in)
    bool order_known = false;           // key_out holds the sorted keys of the live storage order, row_start its rows
    int64_t order_n = 0;                // slots of that order (a slab's arrivals are stored behind them until the next binning)
    uint32_t *key_alt = nullptr;        // takes turns with key_out
    uint32_t *rb_ctl = nullptr;         // PM_RB_WORDS control words, then a look-back word per tile
    uint64_t *rb_mask = nullptr;        // per 64 slots: which hold a mover (written by the drift; allocated with the context)
    uint32_t *rb_before = nullptr;      // per 64 slots: movers in the slots before
    uint32_t *rb_key[2] = {}, *rb_slot[2] = {}, *rb_at = nullptr;       // the movers: new key and slot (unsorted, sorted), where they go in
    unsigned long long *mover_count = nullptr;  // pm_mover_probe::count
    volatile uint32_t *movers_seen = nullptr, *movers_seen_dev = nullptr;      // pinned host word: estimate of the movers of a recent drift
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint32_t *row_start = nullptr;      // G*G + 1 entries: first particle of every mesh row (i,j,:)
    uint32_t *row_unit = nullptr;       // per mesh row: first unit of a crowded row, or none
    unsigned long long *row_live = nullptr;     // whole meshes: a bit per density row, set where the deposit has something to add (k_plan_units)
    bool rows_marked = false;           // the last deposit left its empty rows unwritten and marked them in row_live instead
    // whole meshes: a bit per POTENTIAL row, set where a particle of the last deposit reads the row in its gather (the rows of
    // its cloud's corners and their neighbours along x and y).  A fused step runs the inverse transform along z for those
    // rows only; potential_partial says that the others are stale, and whoever needs the whole mesh (pm_get_mesh, pm_gradient,
    // the energies) has them transformed first (pm_complete_potential): the half spectrum still holds their input.
    unsigned long long *row_need = nullptr;
    bool potential_partial = false;
    bool want_full_potential = false;   // pm_run: the step about to run is followed by an energy sample
    uint32_t *unit_src = nullptr;       // per unit: the mesh row it is a part of
    uint2 *heavy_rows = nullptr;        // density rows with a crowded source: (row, has a partial sum in light_rows)
    uint32_t *counters = nullptr;       // units handed out, density rows listed: two sets of 64 words that take turns
    int counter_set = 0;                // the set the next deposit uses (zeroed by the deposit before it)
    volatile unsigned long long *units_seen_dev = nullptr;    // units_seen (four words) as the device addresses it
    double *unit_rows = nullptr;        // four rows of G doubles per unit
    double *light_rows = nullptr;       // one row of G doubles per listed density row

    // meshes
    float *density = nullptr, *potential = nullptr, *grad[3] = {};
    float2 *spectrum = nullptr;         // (G, G, Ghp) complex64
    float2 *twiddle = nullptr;          // exp(-2 pi i q / G), q < G (column FFT kernels)
    bool fft_custom = false;            // all three axes by the hand-written kernels (no rocFFT plan)
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    void *fft_work = nullptr;
    size_t fft_work_bytes = 0;

    double *density64 = nullptr;        // float64 mesh of the NGP deposit (allocated on first use)
    // the NGP branch in float64 (allocated on first use): potential, its gradient, the half spectrum, rocFFT plans
    double *potential64 = nullptr, *grad64[3] = {nullptr, nullptr, nullptr}, *forces64 = nullptr;
    double2 *spectrum64 = nullptr;
    rocfft_plan fwd64 = nullptr, inv64 = nullptr;
    rocfft_execution_info info64 = nullptr;
    void *fft_work64 = nullptr;
    double *diag = nullptr;             // reduction partials and results of the energy diagnostics
    float *forces = nullptr;            // (n,3) AoS in storage order
    void *stage = nullptr;              // staging for host copies: cap * 32 bytes

    // PM_FLAG_TIMING: per stage and per timed kernel (slots PM_NSTAGES ...) a ring of event pairs around the launches
    static constexpr int TIMED = PM_NSTAGES + PM_NKERNELS, RING = 8;
    hipEvent_t ev[TIMED][RING][2] = {};
    int ev_head[TIMED] = {};            // ring entry the next interval uses
    int ev_pending[TIMED] = {};         // intervals recorded and not read yet (the oldest is entry head - pending)
    float ms[TIMED] = {};
    int calls[TIMED] = {};

    // slab decomposition (pm_create_slab)
    bool slab = false;
    int nslabs = 1;
    // A slab context transforms mesh planes [slab0, slab0 + slabn) -- that never changes -- and holds the particles whose cell
    // lies in a WINDOW of planes [g.i0, g.i0 + g.Gs), which is the slab itself unless pm_slab_set_window said otherwise
    // (contexts made with room for wmax > slabn planes).  The deposit writes the window's planes (spectrum / density), the
    // gather reads the window's potential planes; mesh_spec and mesh_pot are the slab's own z-transformed planes and
    // potential planes, which alias the window's buffers when the context has no room for any other window.
    int slab0 = 0, slabn = 0, wmax = 0;
    float2 *mesh_spec = nullptr;        // slabn planes of the half spectrum (== spectrum when wmax == slabn)
    float *mesh_pot = nullptr;          // slabn planes of the potential (== potential + one plane when wmax == slabn)
    int *cuts = nullptr;                // device: nslabs + 1 first planes of the ranks' windows (cuts[nslabs] = G)
    bool y_in_exchange = false;         // hand-written transforms: the passes along y read / write the all-to-all buffer themselves
    bool slab_check = false;            // the next deposit verifies that every particle lies in the slab
    bool outbox_pending = false;        // a drift has run and its leavers have not been taken out yet (pm_slab_outbox*)
    int64_t leaving = 0;                // particles with the "not mine" key in the live set (they sort to the tail)
    pm_migrant *outbox = nullptr;       // particles that left the slab in the last drift, in no particular order
    uint64_t *outbox_key = nullptr, *outbox_key2 = nullptr;     // (destination slab << 32) | particle id
    uint32_t *outbox_idx2 = nullptr;    // outbox slots in sorted order
    uint32_t *outbox_offsets = nullptr; // first sorted entry per destination slab (nslabs + 1)
    uint32_t *outbox_count = nullptr;   // [0] entries of outbox, [1] entries that did not fit, [2 + d] entries bound for slab d
    int64_t outbox_cap = 0;
    void *outbox_tmp = nullptr;
    size_t outbox_tmp_bytes = 0;
    rocfft_plan slab_fwd2 = nullptr, slab_inv2 = nullptr, slab_fwdx = nullptr, slab_invx = nullptr;

    std::string err;
};

// Times what is enqueued on the context's stream during its lifetime into slot `s` (a PM_STAGE_* value, or PM_NSTAGES +
// a PM_KERNEL_* value) when the context has PM_FLAG_TIMING.  An interval is read (the host waits for its end event) only
// when its ring entry is needed again, RING intervals later, or by pm_timer::flush -- the host does not wait between
// the launches of a step.
struct pm_timer {
    pm_ctx *c; int s; bool on;
    pm_timer(pm_ctx *c_, int s_) : c(c_), s(s_), on(c_ && (c_->flags & PM_FLAG_TIMING) != 0)
    {
        if (!on) return;
        if (c->ev_pending[s] == pm_ctx::RING) read_oldest(c, s);
        if (hipEventRecord(c->ev[s][c->ev_head[s]][0], c->stream) != hipSuccess) on = false;
    }
    ~pm_timer()
    {
        if (!on) return;
        if (hipEventRecord(c->ev[s][c->ev_head[s]][1], c->stream) != hipSuccess) return;    // the interval is dropped
        c->ev_head[s] = (c->ev_head[s] + 1) % pm_ctx::RING;
        c->ev_pending[s] += 1;
    }
    static void read_oldest(pm_ctx *c, int s)
    {
        const int e = (c->ev_head[s] - c->ev_pending[s] + 2 * pm_ctx::RING) % pm_ctx::RING;
        float ms = 0.0f;
        if (hipEventSynchronize(c->ev[s][e][1]) == hipSuccess && hipEventElapsedTime(&ms, c->ev[s][e][0], c->ev[s][e][1]) == hipSuccess) {
            c->ms[s] += ms; c->calls[s] += 1;
        }
        c->ev_pending[s] -= 1;
    }
    static void flush(pm_ctx *c, int s) { while (c->ev_pending[s] > 0) read_oldest(c, s); }
};

size_t pm_heavy_row_entries(int G, int64_t cap);
size_t pm_max_units(int64_t cap);

void pm_set_create_error(const std::string &msg);
// (window_planes: the most planes a slab's particle window may span; nplanes for a context whose window is its slab)
pm_ctx *pm_create_planes(int grid_size, double box_size, int64_t n_capacity, int device, unsigned flags, int plane0, int nplanes, bool slab, int window_planes);

// ---- launchers (each enqueues on ctx->stream and returns a hipError_t) ----------------------
// stage -> records: positions (n,3) of the given dtype + masses (or ones), velocities (or zeros) + iota ids
hipError_t pm_launch_pack(pm_ctx *c, const void *pos_aos, const float *mass, const void *vel_aos);
hipError_t pm_launch_cell_keys(pm_ctx *c);
// synthetic sets drawn on the device into the live float32 records; the number of parameters a kind takes (-1: no such kind)
hipError_t pm_launch_generate(pm_ctx *c, int kind, int64_t n, uint64_t seed, const double *params);
// the same set, of which a slab context keeps the particles whose cell lies in its window (in the order of their numbers):
// counts per block of 256 numbers -> (the caller scans them) -> the records; and the particles per mesh plane of the whole set
hipError_t pm_launch_generate_count(pm_ctx *c, int kind, int64_t n, uint64_t seed, const double *params, uint32_t *block_counts);
hipError_t pm_launch_generate_keep(pm_ctx *c, int kind, int64_t n, uint64_t seed, const double *params, const uint32_t *block_first);
hipError_t pm_launch_generate_planes(const pm_geom &g, hipStream_t stream, int kind, int64_t n, uint64_t seed, const double *params,
                                     unsigned long long *plane_counts);
hipError_t pm_launch_plane_counts(pm_ctx *c, unsigned long long *plane_counts);      // live particles per mesh plane (adds to the G counters)
hipError_t pm_launch_rekey(pm_ctx *c);         // key_in of every live particle under the present window; those outside it go to the outbox
int pm_generate_param_count(int kind);
hipError_t pm_sort_pairs(pm_ctx *c, int bits);          // key_in -> key_out (sorted), src (storage slot of each sorted entry)
// the same from the cell order of the step before (key_out, row_start) and the new keys (key_in); key_out / key_alt swap
hipError_t pm_rebin_incremental(pm_ctx *c, int bits);
// key_in -> key_out, src, by whichever of the two ways the drift's estimate of its movers favours (pm_api.hip)
int pm_bin_order(pm_ctx *c, int bits);
size_t pm_rebin_ctl_bytes(int64_t cap);
size_t pm_sort_counted_tmp_bytes(int64_t n_max);
hipError_t pm_sort_pairs_counted(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n_max,
                                 const uint32_t *n_dev, int bits, int device, hipStream_t s);
hipError_t pm_launch_clear(void *p, size_t bytes, hipStream_t s);      // zero-fill by a kernel (bytes and p multiples of 16)
hipError_t pm_launch_reorder(pm_ctx *c, bool row_starts);
size_t pm_sort_tmp_bytes(int64_t cap);
hipError_t pm_settle_velocities(pm_ctx *c);     // must run before src is overwritten by a new sort
hipError_t pm_launch_row_start(pm_ctx *c);
hipError_t pm_launch_deposit(pm_ctx *c, bool to_spectrum);
hipError_t pm_launch_green(pm_ctx *c);
bool pm_fft_custom_supported(int G);
int pm_fft_spectrum_interleave(int G);       // log2 of the planes interleaved in a whole-mesh half spectrum (pm_geom::sb)
int pm_fft_pitch(int G);
hipError_t pm_fft_upload_twiddles(pm_ctx *c);
hipError_t pm_launch_col_fft(pm_ctx *c, int dir, bool rows_marked = false);
// the same transforms on `planes` planes of a slab: z and y passes between real planes and spectrum planes, and the
// x pass (forward, Green multiply, inverse) on the transposed spectrum (G, planes, Ghp) whose rows are y = y0 ...
hipError_t pm_launch_row_fft_planes(pm_ctx *c, bool forward, const float *real_in, float *real_out, float2 *spec, int planes);
hipError_t pm_launch_col_fft_planes(pm_ctx *c, int dir, float2 *spec, int planes);
hipError_t pm_launch_solve_x(pm_ctx *c, float2 *spec_t, int y0, int ny);
hipError_t pm_launch_col_fft_blocks(pm_ctx *c, int dir, float2 *spec, float2 *blocks, int planes, int first, int count);
hipError_t pm_launch_add(pm_ctx *c, float *mesh, const float *other, int64_t n);
hipError_t pm_launch_green_t(pm_ctx *c, float2 *spec_t, int y0, int ny);
hipError_t pm_launch_transpose_pack(pm_ctx *c, const float2 *spec, float2 *out, bool unpack);
hipError_t pm_sort_outbox(pm_ctx *c, uint32_t count);
size_t pm_sort_outbox_tmp_bytes(int64_t cap);
// 64-bit keys with their indices 0 .. n-1 as values, no context: keys_out sorted, order[r] = where entry r came from
hipError_t pm_sort_u64(void *tmp, size_t tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, uint32_t *order, size_t n, int device, hipStream_t s);
hipError_t pm_launch_pack_outbox(pm_ctx *c, uint32_t count, pm_migrant *dst);
hipError_t pm_launch_outbox_offsets(pm_ctx *c, uint32_t count, uint32_t *offsets);
hipError_t pm_launch_append(pm_ctx *c, const pm_migrant *src, int64_t count);
hipError_t pm_launch_set_ids(pm_ctx *c, const uint32_t *ids);
hipError_t pm_launch_export_slab(pm_ctx *c, uint32_t *ids, float *pos, float *vel);
hipError_t pm_launch_outbox_counts(pm_ctx *c, int64_t *device_counts);
int pm_complete_potential(pm_ctx *c);
// needed_only (inverse only): only the rows that row_need marks
hipError_t pm_launch_row_fft(pm_ctx *c, bool forward, bool needed_only = false);
// (rows_marked: the forward pass along y takes the rows that row_live marks empty as zeros, without reading them)
hipError_t pm_launch_col_fft_tiles(pm_ctx *c, int dir, int tile0, int ntiles, bool rows_marked = false);
int pm_fft_tiles(const pm_ctx *c);
hipError_t pm_launch_gradient(pm_ctx *c);
hipError_t pm_launch_gather(pm_ctx *c, double dt, bool integrate, bool store_forces, bool fused_gradient);
hipError_t pm_launch_export_particles(pm_ctx *c, void *pos_aos, void *vel_aos);
hipError_t pm_launch_export_forces(pm_ctx *c, float *aos);
hipError_t pm_launch_export_forces64(pm_ctx *c, double *aos);
hipError_t pm_launch_export_keys(pm_ctx *c, int64_t *keys, int64_t *perm);
hipError_t pm_launch_energies(pm_ctx *c, double *d_ke, double *d_pe);
hipError_t pm_noalloc_prepare(int device);    // looks up the device address of pm_energies_noalloc's static scratch
hipError_t pm_launch_force_ngp(pm_ctx *c);
hipError_t pm_launch_ngp_keys(pm_ctx *c, uint32_t *bad);
hipError_t pm_launch_ngp_sum(pm_ctx *c);
hipError_t pm_launch_green64(pm_ctx *c);
hipError_t pm_launch_gradient64(pm_ctx *c);
hipError_t pm_launch_ngp_gather(pm_ctx *c, double dt, bool integrate, double *forces);
hipError_t pm_launch_deposit64(pm_ctx *c);       // cloud-in-cell in float64 (PM_FLAG_F64): density64 from pd[], forces / kick / drift on grad64
hipError_t pm_launch_gather64(pm_ctx *c, double dt, bool integrate, double *forces);
