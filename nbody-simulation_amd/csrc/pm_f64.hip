// The cloud-in-cell branch with every array in float64 (PM_FLAG_F64): what main.py:50-60 computes when the mesh of
// functions.py:22, the forces of functions.py:102 and the positions after main.py:60 are float64 instead of float32.
// The reference has no such switch; this is the build mode that is compared with a float64 restatement at 1e-12, where
// the float32 path can only be held to 1e-5.  Deposit and gather are kernels of their own here, written for that
// comparison and not tuned: the solve and the gradient are the float64 chain of the nearest-grid-point branch
// (pm_potential_f64, pm_gradient_f64).
//
// Arithmetic as in pm_particles.hip: wrap, scale, fractions and weights in float64 (functions.py:27-51), corner order
// 000 100 010 001 110 101 011 111 (functions.py:61-68, 140-178); built with -ffp-contract=off.
#include "pm_internal.hpp"
#include "pm_cell_dev.hpp"

namespace {

constexpr int BLOCK = 256;

using pmcell::cell_t;
using pmcell::locate;
using pmcell::cell_key;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

// np.remainder(x, L) for L > 0 in float64 (main.py:60 without its cast): may return exactly L.
__device__ __forceinline__ double wrap_np(double x, double L)
{
    if (x >= 0.0 && x < L) return (x == 0.0) ? 0.0 : x;
    double r = fmod(x, L);
    if (r != 0.0) {
        if (r < 0.0) r += L;
    } else {
        r = 0.0;
    }
    return r;
}

// One wave per density row (i, j, :): the particles of the four mesh rows (i - di, j - dj) -- contiguous runs of the
// particle array in cell order -- add mass * wx * wy * wz (left to right, functions.py:61-68) for their corners
// (di, dj, 0) and (di, dj, 1) into a float64 copy of the row in LDS, which then leaves as it is.
__global__ void __launch_bounds__(BLOCK)
k_deposit64(const pm_prec<double> *__restrict__ P, const uint32_t *__restrict__ rs, double *__restrict__ density, pm_geom g)
{
    extern __shared__ double rows[];                    // BLOCK / 64 rows of G doubles
    const int G = g.G, lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const uint32_t row = blockIdx.x * (BLOCK / 64) + (uint32_t)w;
    if (row >= (uint32_t)G * (uint32_t)G) return;
    double *acc = rows + (size_t)w * G;
    for (int c = lane; c < G; c += 64) acc[c] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int i = (int)(row / (uint32_t)G), j = (int)(row % (uint32_t)G);
    const int im = i == 0 ? G - 1 : i - 1, jm = j == 0 ? G - 1 : j - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool di = q & 1, dj = q & 2;
        const uint32_t src = (uint32_t)((di ? im : i) * G + (dj ? jm : j));
        const uint32_t beg = rs[src], end = rs[src + 1];
        for (uint32_t r = beg + (uint32_t)lane; r < end; r += 64) {
            const pm_prec<double> p = P[r];
            const cell_t c = locate(p, g);
            const double m = (double)p.m * (di ? c.dx : 1.0 - c.dx) * (dj ? c.dy : 1.0 - c.dy);
            __hip_atomic_fetch_add(&acc[c.k], m * (1.0 - c.dz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&acc[c.k + 1 == G ? 0 : c.k + 1], m * c.dz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double *out = density + (size_t)row * G;
    for (int c = lane; c < G; c += 64) out[c] = acc[c];
}

__device__ constexpr int DI[8] = {0, 1, 0, 0, 1, 1, 0, 1}, DJ[8] = {0, 0, 1, 0, 1, 0, 1, 1}, DK[8] = {0, 0, 0, 1, 0, 1, 1, 1};

// force_CIC (functions.py:99-180) on the float64 gradient, forces accumulated in float64; then main.py:58-60 in float64
// (velocities += forces * dt; positions += velocities * dt; positions % box_size) and the cell key of the new position.
// MAPPED: the velocity record of slot r is record vmap[r] of Vin (the records lag one reorder behind).
template <bool MAPPED>
__global__ void __launch_bounds__(BLOCK)
k_gather64(pm_prec<double> *__restrict__ P, const pm_vrec<double> *__restrict__ Vin, pm_vrec<double> *__restrict__ Vout,
           const uint32_t *__restrict__ vmap, const double *__restrict__ gx, const double *__restrict__ gy, const double *__restrict__ gz,
           double *__restrict__ forces, uint32_t *__restrict__ key_next, int64_t n, pm_geom g, double dt, bool integrate)
{
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const int G = g.G;
    const pm_prec<double> p = P[r];
    const cell_t c = locate(p, g);
    const int i1 = (c.i + 1 == G) ? 0 : c.i + 1, j1 = (c.j + 1 == G) ? 0 : c.j + 1, k1 = (c.k + 1 == G) ? 0 : c.k + 1;
    const double wx[2] = {1.0 - c.dx, c.dx}, wy[2] = {1.0 - c.dy, c.dy}, wz[2] = {1.0 - c.dz, c.dz};
    const int ii[2] = {c.i, i1}, jj[2] = {c.j, j1}, kk[2] = {c.k, k1};
    double fx = 0.0, fy = 0.0, fz = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int64_t node = ((int64_t)ii[DI[q]] * G + jj[DJ[q]]) * G + kk[DK[q]];
        const double w = wx[DI[q]] * wy[DJ[q]] * wz[DK[q]];
        fx = fx + gx[node] * w; fy = fy + gy[node] * w; fz = fz + gz[node] * w;
    }
    if (forces) { forces[3 * r] = fx; forces[3 * r + 1] = fy; forces[3 * r + 2] = fz; }
    if (!integrate) return;
    pm_vrec<double> v = Vin[MAPPED ? vmap[r] : r];
    v.x = v.x + fx * dt; v.y = v.y + fy * dt; v.z = v.z + fz * dt;
    Vout[r] = v;
    pm_prec<double> o = p;
    o.x = wrap_np(p.x + v.x * dt, g.L);
    o.y = wrap_np(p.y + v.y * dt, g.L);
    o.z = wrap_np(p.z + v.z * dt, g.L);
    P[r] = o;
    key_next[r] = cell_key(locate(o, g), g);
}

}  // namespace

// density64 <- CIC of the live float64 positions (which must be in cell order, row_start matching)
hipError_t pm_launch_deposit64(pm_ctx *c)
{
    const int G = c->g.G;
    const uint32_t rows = (uint32_t)G * (uint32_t)G;
    const size_t lds = (size_t)(BLOCK / 64) * G * sizeof(double);
    if (lds > 64 * 1024) return hipErrorInvalidValue;     // (grid sizes beyond 2048: not in this mode)
    k_deposit64<<<dim3((rows + BLOCK / 64 - 1) / (BLOCK / 64)), BLOCK, lds, c->stream>>>(c->pd[c->cur], c->row_start, c->density64, c->g);
    return hipGetLastError();
}

// forces != nullptr: (n,3) float64 forces in storage order; integrate: kick, drift, wrap and the next cell keys
hipError_t pm_launch_gather64(pm_ctx *c, double dt, bool integrate, double *forces)
{
    if (c->n == 0) return hipSuccess;
    const pm_vrec<double> *vin = c->vd[c->vcur];
    pm_vrec<double> *vout = c->v_stale ? c->vd[1 - c->vcur] : c->vd[c->vcur];
    const double *gx = c->grad64[0], *gy = c->grad64[1], *gz = c->grad64[2];
    if (c->v_stale)
        k_gather64<true><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pd[c->cur], vin, vout, c->src, gx, gy, gz, forces, c->key_in, c->n, c->g, dt, integrate);
    else
        k_gather64<false><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pd[c->cur], vin, vout, nullptr, gx, gy, gz, forces, c->key_in, c->n, c->g, dt, integrate);
    if (integrate && c->v_stale) { c->vcur = 1 - c->vcur; c->v_stale = false; }
    return hipGetLastError();
}
