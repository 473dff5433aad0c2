// Diagnostics and the nearest-grid-point variants of the reference, for gfx950:
// kinetic / potential energy (functions.py:352-361), force_NGP (functions.py:92-97) and the NGP
// deposit (functions.py:11-16).  Built with -ffp-contract=off: the index formulas below must round
// exactly like NumPy evaluates them in the positions' dtype.
#include "pm_internal.hpp"

#include <mutex>

namespace {

constexpr int BLOCK = 256;
constexpr int RED_BLOCKS = 1024;

// Scratch of pm_energies_noalloc: the per-block partial sums of the two reductions (kinetic, potential), 16 KiB that
// every GPU the library runs on holds until the process ends.  In host code the array's name only stands for the
// symbol; the address the device uses comes from hipGetSymbolAddress, once per device (noalloc_scratch below).
static __device__ __align__(256) double scratch[2 * RED_BLOCKS];

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

// (pos / box_size * grid_size).astype(int) % grid_size, evaluated in the positions' dtype
// (functions.py:94 and :358); Python's % gives a result in [0, G).
template <typename T>
__device__ __forceinline__ int ngp_wrapped_index(T x, T L, T Gf, int G)
{
    T s = (x / L) * Gf;
    long long t = (long long)s;                 // astype(int) truncates toward zero
    int r = (int)(t % G);
    return r < 0 ? r + G : r;
}

// np.remainder(x, L) for L > 0 in the positions' dtype (functions.py:13): may return exactly L.
template <typename T> __device__ __forceinline__ T np_remainder(T x, T L)
{
    T r = (sizeof(T) == 4) ? (T)fmodf((float)x, (float)L) : (T)fmod((double)x, (double)L);
    if (r != (T)0) {
        if (r < (T)0) r += L;
    } else {
        r = (T)0;
    }
    return r;
}

// Deterministic sum: every thread adds its grid-stride sequence in order, a fixed butterfly and a
// fixed order over the waves give the block's partial; k_reduce_final adds the partials in order.
__device__ __forceinline__ void block_partial(double v, double *__restrict__ partial)
{
    __shared__ double wsum[BLOCK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) s += wsum[w];
        partial[blockIdx.x] = s;
    }
}

// KE = 0.5 * sum(m * (vx^2 + vy^2 + vz^2)): the per-particle term in the dtype NumPy gives it
// (float32 for float32 velocities), the sum in float64.
template <typename TV>
__global__ void __launch_bounds__(BLOCK)
k_kinetic(const pm_prec<float> *__restrict__ Pf, const pm_prec<double> *__restrict__ Pd, const pm_vrec<TV> *__restrict__ V,
          const uint32_t *__restrict__ vmap, int64_t n, double *__restrict__ partial)
{
    double acc = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * BLOCK) {
        const float m = Pf ? Pf[r].m : Pd[r].m;
        const pm_vrec<TV> v = V[vmap ? vmap[r] : r];
        TV v2 = (v.x * v.x + v.y * v.y) + v.z * v.z;
        acc += (sizeof(TV) == 4) ? (double)(m * (float)v2) : (double)m * (double)v2;
    }
    block_partial(acc, partial);
}

// PE = 0.5 * sum(m * phi[ngp cell]): float32 product, float64 sum.
// (with a float64 potential -- the NGP branch -- the product is float64 as well)
template <typename TP, typename TM>
__global__ void __launch_bounds__(BLOCK)
k_potential_energy(const pm_prec<TP> *__restrict__ P, const TM *__restrict__ phi, int64_t n, pm_geom g, double *__restrict__ partial)
{
    const int G = g.G;
    const TP L = (TP)g.L, Gf = (TP)G;
    double acc = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * BLOCK) {
        const pm_prec<TP> p = P[r];
        int i = ngp_wrapped_index<TP>(p.x, L, Gf, G), j = ngp_wrapped_index<TP>(p.y, L, Gf, G), k = ngp_wrapped_index<TP>(p.z, L, Gf, G);
        acc += (double)((TM)p.m * phi[((int64_t)i * G + j) * G + k]);
    }
    block_partial(acc, partial);
}

__global__ void k_reduce_final(const double *__restrict__ partial, int count, double scale, double *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int q = 0; q < count; ++q) s += partial[q];
        *out = s * scale;
    }
}

// force_NGP: the three gradient meshes at the particle's nearest-grid-point cell.
template <typename TP>
__global__ void k_force_ngp(const pm_prec<TP> *__restrict__ P, const float *__restrict__ gx, const float *__restrict__ gy,
                            const float *__restrict__ gz, float *__restrict__ forces, int64_t n, pm_geom g)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const int G = g.G;
    const TP L = (TP)g.L, Gf = (TP)G;
    const pm_prec<TP> p = P[r];
    int i = ngp_wrapped_index<TP>(p.x, L, Gf, G), j = ngp_wrapped_index<TP>(p.y, L, Gf, G), k = ngp_wrapped_index<TP>(p.z, L, Gf, G);
    int64_t q = ((int64_t)i * G + j) * G + k;
    forces[3 * r] = gx[q]; forces[3 * r + 1] = gy[q]; forces[3 * r + 2] = gz[q];
}

// NGP deposit keys: floor((pos % L) / L * G) per axis in the positions' dtype (functions.py:13-14).
// An index that reaches G (the float32 remainder can return exactly L) is out of range in the
// reference (IndexError): flagged, and folded to G-1 so the sort stays in range.
template <typename TP>
__global__ void k_ngp_keys(const pm_prec<TP> *__restrict__ P, uint32_t *__restrict__ key, int64_t n, pm_geom g, uint32_t *__restrict__ bad)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const int G = g.G;
    const TP L = (TP)g.L, Gf = (TP)G;
    const pm_prec<TP> p = P[r];
    TP c[3] = {p.x, p.y, p.z};
    int idx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        TP s = (np_remainder<TP>(c[a], L) / L) * Gf;
        int t = (int)floor((double)s);
        if (t >= G || t < 0) { atomicOr(bad, 1u); t = t < 0 ? 0 : G - 1; }
        idx[a] = t;
    }
    key[r] = (uint32_t)(((int64_t)idx[0] * G + idx[1]) * G + idx[2]);
}

// np.add.at(density, cell, masses) with a float64 mesh: after a stable sort by cell the masses of a
// cell are one run (slot src[r] of the particle array); the thread of the run's first element adds
// them in order, runs of more than 64 by its whole wave (stride-64 partial sums, fixed butterfly).
template <typename TP>
__global__ void __launch_bounds__(BLOCK)
k_ngp_sum(const pm_prec<TP> *__restrict__ P, const uint32_t *__restrict__ key, const uint32_t *__restrict__ src, int64_t n,
          double *__restrict__ density)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    bool head = r < n && (r == 0 || key[r - 1] != key[r]);
    int64_t end = r;
    uint32_t k0 = head ? key[r] : 0u;
    if (head) {
        // first index past the run: gallop, then bisect
        int64_t inside = r, step = 1, outside = n;
        while (inside + step < n) {
            if (key[inside + step] == k0) { inside += step; step <<= 1; }
            else { outside = inside + step; break; }
        }
        while (outside - inside > 1) {
            int64_t mid = (inside + outside) >> 1;
            if (key[mid] == k0) inside = mid; else outside = mid;
        }
        end = outside;
    }
    int64_t len = end - r;
    if (head && len <= 64) {
        double s = 0.0;
        for (int64_t q = r; q < end; ++q) s += (double)P[src[q]].m;
        density[k0] = s;
    }
    unsigned long long big = __ballot(head && len > 64);
    while (big) {
        int l = __ffsll((long long)big) - 1;
        big &= big - 1;
        int64_t b = __shfl(r, l), e = __shfl(end, l);
        double part = 0.0;
        for (int64_t q = b + lane_id(); q < e; q += 64) part += (double)P[src[q]].m;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (lane_id() == l) density[k0] = part;
    }
}


// ---- the NGP branch end to end in float64 -----------------------------------------------------------
// NGP returns a float64 mesh (functions.py:12), so compute_potential (functions.py:73-89) and np.gradient
// (main.py:54) run in float64 for that branch and force_NGP (functions.py:92-97) returns float64 forces.

// np.fft.fftfreq(G)[a] as float32 (functions.py:75)
__device__ __forceinline__ float freq32(int a, int G, double invG)
{
    int f = (a < (G + 1) / 2) ? a : a - G;
    return (float)((double)f * invG);
}

// potential_k = density_k / k2 (functions.py:85): k2 = kx**2 + ky**2 + kz**2 in float32, and NumPy divides a
// complex128 by it as a multiplication by 1.0 / k2; the 1/M of ifftn (functions.py:89) goes in here too.
__global__ void k_green64(double2 *__restrict__ S, int G, int Gh, double invM, double soft)
{
    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= (int64_t)G * G * Gh) return;
    const int c = (int)(q % Gh), j = (int)((q / Gh) % G), i = (int)(q / ((int64_t)Gh * G));
    const double invG = 1.0 / (double)G;
    const float kx = freq32(i, G, invG), ky = freq32(j, G, invG), kz = freq32(c, G, invG);
    const float k2 = __fadd_rn(__fadd_rn(__fmul_rn(kx, kx), __fmul_rn(ky, ky)), __fmul_rn(kz, kz));
    double2 v = S[q];
    if ((i | j | c) == 0) v = make_double2(0.0, 0.0);                   // functions.py:86
    else {
        double r = 1.0 / (double)k2;
        if (soft != 0.0) r *= exp(-soft * (double)k2);
        v.x = v.x * r * invM; v.y = v.y * r * invM;
    }
    S[q] = v;
}

// np.gradient(f) in float64: (f[n+1] - f[n-1]) / 2 inside, one-sided differences on the faces
__global__ void k_gradient64(const double *__restrict__ f, double *__restrict__ gx, double *__restrict__ gy, double *__restrict__ gz, int G)
{
    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= (int64_t)G * G * G) return;
    const int k = (int)(q % G), j = (int)((q / G) % G), i = (int)(q / ((int64_t)G * G));
    auto d = [&](int a, int64_t stride) {
        if (a == 0) return f[q + stride] - f[q];
        if (a == G - 1) return f[q] - f[q - stride];
        return (f[q + stride] - f[q - stride]) * 0.5;
    };
    gx[q] = d(i, (int64_t)G * G); gy[q] = d(j, G); gz[q] = d(k, 1);
}

// force_NGP on the float64 gradient and, with integrate, main.py:58-60 behind it: `velocities += forces * dt` has
// a float64 right-hand side (a float32 velocity is widened for the add and rounded once), the drift and the wrap
// are those of the cloud-in-cell branch.
template <typename TP, typename TV>
__global__ void __launch_bounds__(BLOCK)
k_ngp_gather(const pm_prec<TP> *__restrict__ Pin, pm_prec<float> *__restrict__ Pout, pm_vrec<TV> *__restrict__ V,
             const double *__restrict__ gx, const double *__restrict__ gy, const double *__restrict__ gz,
             double *__restrict__ forces, int64_t n, pm_geom g, double dt, bool integrate)
{
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const int G = g.G;
    const TP L = (TP)g.L, Gf = (TP)G;
    const pm_prec<TP> p = Pin[r];
    const int i = ngp_wrapped_index<TP>(p.x, L, Gf, G), j = ngp_wrapped_index<TP>(p.y, L, Gf, G), k = ngp_wrapped_index<TP>(p.z, L, Gf, G);
    const int64_t q = ((int64_t)i * G + j) * G + k;
    const double fx = gx[q], fy = gy[q], fz = gz[q];
    if (forces) { forces[3 * r] = fx; forces[3 * r + 1] = fy; forces[3 * r + 2] = fz; }
    if (!integrate) return;
    pm_vrec<TV> v = V[r];
    v.x = (TV)((double)v.x + fx * dt); v.y = (TV)((double)v.y + fy * dt); v.z = (TV)((double)v.z + fz * dt);
    V[r] = v;
    pm_prec<float> o;
    o.m = p.m;
    const float Lf = g.Lf;
    auto wrap = [&](float x) {                       // np.remainder(x, L) in float32: may return exactly L
        if (x >= 0.0f && x < Lf) return (x == 0.0f) ? 0.0f : x;
        return np_remainder<float>(x, Lf);
    };
    if constexpr (sizeof(TV) == 4 && sizeof(TP) == 4) {
        const float dtf = (float)dt;
        o.x = wrap(p.x + v.x * dtf); o.y = wrap(p.y + v.y * dtf); o.z = wrap(p.z + v.z * dtf);
    } else {
        double mx, my, mz;
        if constexpr (sizeof(TV) == 4) { const float dtf = (float)dt; mx = (double)(v.x * dtf); my = (double)(v.y * dtf); mz = (double)(v.z * dtf); }
        else { mx = v.x * dt; my = v.y * dt; mz = v.z * dt; }
        o.x = wrap((float)((double)p.x + mx)); o.y = wrap((float)((double)p.y + my)); o.z = wrap((float)((double)p.z + mz));
    }
    Pout[r] = o;
}

}  // namespace

// Device address of `scratch` on `device`, looked up on first use with that device current; every GPU has its own
// copy of the array at its own address.
static hipError_t noalloc_scratch(int device, double **out)
{
    constexpr int MAX_DEVICES = 64;
    static std::mutex lock;
    static double *address[MAX_DEVICES] = {};
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(lock);
    if (!address[device]) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        double *p = nullptr;
        e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        address[device] = p;
    }
    *out = address[device];
    return hipSuccess;
}

hipError_t pm_noalloc_prepare(int device)
{
    double *p = nullptr;
    return noalloc_scratch(device, &p);
}

static hipError_t launch_energies(pm_ctx *c, double *partials, double *d_ke, double *d_pe);

// ke and pe are two doubles in device memory (c->diag + 2*RED_BLOCKS ...); the caller copies them out.
hipError_t pm_launch_energies(pm_ctx *c, double *d_ke, double *d_pe)
{
    return launch_energies(c, c->diag, d_ke, d_pe);
}

// partials: 2 * RED_BLOCKS doubles.  Every launched block writes its own entry and k_reduce_final reads exactly the
// entries of the launched blocks, so nothing in `partials` has to start at zero.
static hipError_t launch_energies(pm_ctx *c, double *partials, double *d_ke, double *d_pe)
{
    const int64_t n = c->n;
    double *part_k = partials, *part_p = partials + RED_BLOCKS;
    int blocks = (int)((n + BLOCK - 1) / BLOCK);
    if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
    if (blocks < 1) blocks = 1;
    const int u = c->cur;
    const uint32_t *vmap = c->v_stale ? c->src : nullptr;
    const pm_prec<float> *pf = c->pos_f64 ? nullptr : c->pf[u];
    const pm_prec<double> *pd = c->pos_f64 ? c->pd[u] : nullptr;
    if (c->vel_f64) k_kinetic<double><<<blocks, BLOCK, 0, c->stream>>>(pf, pd, c->vd[c->vcur], vmap, n, part_k);
    else k_kinetic<float><<<blocks, BLOCK, 0, c->stream>>>(pf, pd, c->vf[c->vcur], vmap, n, part_k);
    if (c->flags & (PM_FLAG_NGP | PM_FLAG_F64)) {
        if (c->pos_f64) k_potential_energy<double, double><<<blocks, BLOCK, 0, c->stream>>>(c->pd[u], c->potential64, n, c->g, part_p);
        else k_potential_energy<float, double><<<blocks, BLOCK, 0, c->stream>>>(c->pf[u], c->potential64, n, c->g, part_p);
    } else {
        if (c->pos_f64) k_potential_energy<double, float><<<blocks, BLOCK, 0, c->stream>>>(c->pd[u], c->potential, n, c->g, part_p);
        else k_potential_energy<float, float><<<blocks, BLOCK, 0, c->stream>>>(c->pf[u], c->potential, n, c->g, part_p);
    }
    k_reduce_final<<<1, 64, 0, c->stream>>>(part_k, blocks, 0.5, d_ke);
    k_reduce_final<<<1, 64, 0, c->stream>>>(part_p, blocks, 0.5, d_pe);
    return hipGetLastError();
}

hipError_t pm_launch_force_ngp(pm_ctx *c)
{
    if (c->n == 0) return hipSuccess;
    const int u = c->cur;
    if (c->pos_f64) k_force_ngp<double><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pd[u], c->grad[0], c->grad[1], c->grad[2], c->forces, c->n, c->g);
    else k_force_ngp<float><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pf[u], c->grad[0], c->grad[1], c->grad[2], c->forces, c->n, c->g);
    return hipGetLastError();
}

// keys -> key_in; `bad` is one device word, set when an index is out of range
hipError_t pm_launch_ngp_keys(pm_ctx *c, uint32_t *bad)
{
    if (c->n == 0) return hipSuccess;
    const int u = c->cur;
    if (c->pos_f64) k_ngp_keys<double><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pd[u], c->key_in, c->n, c->g, bad);
    else k_ngp_keys<float><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pf[u], c->key_in, c->n, c->g, bad);
    return hipGetLastError();
}

// after pm_sort_pairs: key_out sorted, src = storage slot of each sorted element
hipError_t pm_launch_ngp_sum(pm_ctx *c)
{
    if (c->n == 0) return hipSuccess;
    const int u = c->cur;
    if (c->pos_f64) k_ngp_sum<double><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pd[u], c->key_out, c->src, c->n, c->density64);
    else k_ngp_sum<float><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pf[u], c->key_out, c->src, c->n, c->density64);
    return hipGetLastError();
}

// ---- float64 NGP branch: launchers ------------------------------------------------------------------

hipError_t pm_launch_green64(pm_ctx *c)
{
    const int G = c->g.G, Gh = c->g.Gh;
    k_green64<<<grid_for((int64_t)G * G * Gh), BLOCK, 0, c->stream>>>(c->spectrum64, G, Gh, 1.0 / (double)c->g.M, (double)c->g.soft);
    return hipGetLastError();
}

hipError_t pm_launch_gradient64(pm_ctx *c)
{
    k_gradient64<<<grid_for(c->g.M), BLOCK, 0, c->stream>>>(c->potential64, c->grad64[0], c->grad64[1], c->grad64[2], c->g.G);
    return hipGetLastError();
}

// velocity records must be in storage order (pm_settle_velocities)
hipError_t pm_launch_ngp_gather(pm_ctx *c, double dt, bool integrate, double *forces)
{
    if (c->n == 0) return hipSuccess;
    const int u = c->cur, v = c->vcur;
    const dim3 grid = grid_for(c->n);
    const double *gx = c->grad64[0], *gy = c->grad64[1], *gz = c->grad64[2];
    if (c->pos_f64) {
        if (c->vel_f64) k_ngp_gather<double, double><<<grid, BLOCK, 0, c->stream>>>(c->pd[u], c->pf[u], c->vd[v], gx, gy, gz, forces, c->n, c->g, dt, integrate);
        else k_ngp_gather<double, float><<<grid, BLOCK, 0, c->stream>>>(c->pd[u], c->pf[u], c->vf[v], gx, gy, gz, forces, c->n, c->g, dt, integrate);
    } else {
        if (c->vel_f64) k_ngp_gather<float, double><<<grid, BLOCK, 0, c->stream>>>(c->pf[u], c->pf[u], c->vd[v], gx, gy, gz, forces, c->n, c->g, dt, integrate);
        else k_ngp_gather<float, float><<<grid, BLOCK, 0, c->stream>>>(c->pf[u], c->pf[u], c->vf[v], gx, gy, gz, forces, c->n, c->g, dt, integrate);
    }
    return hipGetLastError();
}

// ---- the energies without any allocation --------------------------------------------------------------

extern "C" int pm_energies_noalloc(pm_ctx *c, double *device_ke_pe)
{
    if (!c) return PM_EINVAL;
    auto fail = [&](int code, const char *msg) { c->err = msg; return code; };
    if (c->slab) return fail(PM_ESTATE, "pm_energies_noalloc works on the whole mesh: not on a slab context");
    if (!c->pf[0]) return fail(PM_ESTATE, "no particles: call pm_set_particles first");
    if (!device_ke_pe) return fail(PM_EINVAL, "device_ke_pe is NULL");
    if (c->n > PM_NOALLOC_MAX_PARTICLES)
        return fail(PM_EINVAL, "pm_energies_noalloc takes at most PM_NOALLOC_MAX_PARTICLES (268435456) particles: use pm_energies");
    if ((c->flags & (PM_FLAG_NGP | PM_FLAG_F64)) && !c->potential64)
        return fail(PM_ESTATE, "no float64 potential yet: step, or call pm_potential_f64 / pm_set_mesh_f64 first");
    if (!(c->flags & (PM_FLAG_NGP | PM_FLAG_F64)) && !(c->mesh_valid & (1u << PM_MESH_POTENTIAL)))
        return fail(PM_ESTATE, "no potential yet: step, or call pm_potential / pm_set_mesh(PM_MESH_POTENTIAL) first");
    if (c->potential_partial && pm_complete_potential(c) != PM_OK) return PM_EHIP;      // (a kernel launch: no allocation, no wait)
    hipError_t e = hipSetDevice(c->device);
    double *partials = nullptr;
    if (e == hipSuccess) e = noalloc_scratch(c->device, &partials);
    if (e == hipSuccess) e = launch_energies(c, partials, device_ke_pe, device_ke_pe + 1);
    if (e != hipSuccess) { c->err = std::string("pm_energies_noalloc: ") + hipGetErrorString(e); return PM_EHIP; }
    return PM_OK;
}
