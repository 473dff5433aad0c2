// Particle-side kernels of the PM step for gfx950: cell keys, reorder into
// cell order, cell-start table, CIC deposit (cell-centric, no float atomics),
// CIC gather fused with kick / drift / wrap, and the host-copy helpers.
//
// Arithmetic follows what numba compiles the reference loops to
// (functions.py:19-70, 99-180): wrap, scale, fractional parts and weights in
// float64; this file is built with -ffp-contract=off so no product is fused
// into an add.
#include "pm_internal.hpp"

namespace {

constexpr int BLOCK = 256;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Python-style x % L for L > 0 (functions.py:27-29): result in [0, L].
__device__ __forceinline__ double wrap_f64(double x, double L)
{
    if (x >= 0.0 && x < L) return (x == 0.0) ? 0.0 : x;    // already inside: fmod is the identity
    double r = fmod(x, L);
    if (r != 0.0) {
        if (r < 0.0) r += L;
    } else {
        r = 0.0;
    }
    return r;
}

// float32 np.remainder(x, L) for L > 0 (main.py:60): may return exactly L.
__device__ __forceinline__ float wrap_f32(float x, float L)
{
    if (x >= 0.0f && x < L) return (x == 0.0f) ? 0.0f : x;
    float r = fmodf(x, L);
    if (r != 0.0f) {
        if (r < 0.0f) r += L;
    } else {
        r = 0.0f;
    }
    return r;
}

struct cell_t {
    int i, j, k;
    double dx, dy, dz;
};

// functions.py:27-43.  A scaled coordinate that rounds up to G (the reference
// would index out of bounds there) is folded back to cell 0 of that axis.
__device__ __forceinline__ cell_t locate(double px, double py, double pz, const pm_geom &g)
{
    cell_t c;
    double sx = wrap_f64(px, g.L) * g.inv;
    double sy = wrap_f64(py, g.L) * g.inv;
    double sz = wrap_f64(pz, g.L) * g.inv;
    c.i = (int)sx; c.j = (int)sy; c.k = (int)sz;
    c.dx = sx - (double)c.i; c.dy = sy - (double)c.j; c.dz = sz - (double)c.k;
    if (c.i >= g.G) c.i -= g.G;
    if (c.j >= g.G) c.j -= g.G;
    if (c.k >= g.G) c.k -= g.G;
    return c;
}

// ---- host copies ---------------------------------------------------------------------------

template <typename TI, typename TO>
__global__ void k_aos_to_soa(const TI *__restrict__ aos, TO *__restrict__ a, TO *__restrict__ b, TO *__restrict__ c, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n) return;
    a[p] = (TO)aos[3 * p]; b[p] = (TO)aos[3 * p + 1]; c[p] = (TO)aos[3 * p + 2];
}

template <typename T>
__global__ void k_fill(T *__restrict__ a, T v, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p < n) a[p] = v;
}

__global__ void k_iota(uint32_t *__restrict__ a, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p < n) a[p] = (uint32_t)p;
}

template <typename TI, typename TO>
__global__ void k_soa_to_aos_by_id(const TI *__restrict__ a, const TI *__restrict__ b, const TI *__restrict__ c,
                                   const uint32_t *__restrict__ id, TO *__restrict__ aos, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    int64_t o = 3 * (int64_t)id[r];
    aos[o] = (TO)a[r]; aos[o + 1] = (TO)b[r]; aos[o + 2] = (TO)c[r];
}

__global__ void k_aos_by_id(const float *__restrict__ in, const uint32_t *__restrict__ id, float *__restrict__ out, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    int64_t o = 3 * (int64_t)id[r];
    out[o] = in[3 * r]; out[o + 1] = in[3 * r + 1]; out[o + 2] = in[3 * r + 2];
}

__global__ void k_export_keys(const uint32_t *__restrict__ key, const uint32_t *__restrict__ id,
                              int64_t *__restrict__ keys, int64_t *__restrict__ perm, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    keys[r] = (int64_t)key[r];
    perm[r] = (int64_t)id[r];
}

// ---- binning -------------------------------------------------------------------------------

template <typename TP>
__global__ void k_cell_keys(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z,
                            uint32_t *__restrict__ key, int64_t n, pm_geom g)
{
    int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n) return;
    cell_t c = locate((double)x[p], (double)y[p], (double)z[p], g);
    key[p] = (uint32_t)(((int64_t)c.i * g.G + c.j) * g.G + c.k);
}

template <typename T>
__device__ __forceinline__ void move3(T *const dst[3], T *const src[3], int64_t r, int64_t s)
{
    dst[0][r] = src[0][s]; dst[1][r] = src[1][s]; dst[2][r] = src[2][s];
}

struct reorder_args {
    float *x_in[3], *x_out[3];
    double *xd_in[3], *xd_out[3];
    float *vf_in[3], *vf_out[3];
    double *vd_in[3], *vd_out[3];
    float *m_in, *m_out;
    uint32_t *id_in, *id_out;
    const uint32_t *src;
    int64_t n;
    bool pos_f64, vel_f64;
};

__global__ void k_reorder(reorder_args a)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= a.n) return;
    int64_t s = a.src[r];
    if (a.pos_f64) move3(a.xd_out, a.xd_in, r, s); else move3(a.x_out, a.x_in, r, s);
    if (a.vel_f64) move3(a.vd_out, a.vd_in, r, s); else move3(a.vf_out, a.vf_in, r, s);
    a.m_out[r] = a.m_in[s];
    a.id_out[r] = a.id_in[s];
}

// cell_start[c] = number of particles whose key is < c, for c in [0, M]: thread r
// (0..n) owns the cells in (key[r-1], key[r]].  Short runs of empty cells are
// written by the owning lane, medium ones by its wave, long ones go to a list
// that k_fill_gaps spreads over the whole grid.
constexpr uint32_t GAP_LANE = 64, GAP_WAVE = 65536;

__global__ void k_cell_start(const uint32_t *__restrict__ key, int64_t n, int64_t M, uint32_t *__restrict__ cs,
                             uint32_t *__restrict__ gap_list, uint32_t *__restrict__ gap_count, uint32_t gap_cap)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    int64_t lo = 0, hi = 0;
    if (r <= n) {
        lo = (r == 0) ? 0 : (int64_t)key[r - 1] + 1;
        hi = (r == n) ? M + 1 : (int64_t)key[r] + 1;
    }
    int64_t len = hi - lo;
    if (len > 0 && len <= GAP_LANE)
        for (int64_t c = lo; c < hi; ++c) cs[c] = (uint32_t)r;
    if (len > GAP_WAVE) {
        uint32_t e = atomicAdd(gap_count, 1u);
        if (e < gap_cap) {      // cannot overflow: at most M / GAP_WAVE + 1 such gaps exist
            gap_list[3 * e] = (uint32_t)lo; gap_list[3 * e + 1] = (uint32_t)(hi - 1); gap_list[3 * e + 2] = (uint32_t)r;
        }
    }
    unsigned long long mid = __ballot(len > GAP_LANE && len <= GAP_WAVE);
    while (mid) {
        int l = __ffsll((long long)mid) - 1;
        mid &= mid - 1;
        int64_t a = __shfl(lo, l), b = __shfl(hi, l);
        uint32_t v = (uint32_t)__shfl(r, l);
        for (int64_t c = a + lane_id(); c < b; c += 64) cs[c] = v;
    }
}

__global__ void k_fill_gaps(const uint32_t *__restrict__ gap_list, const uint32_t *__restrict__ gap_count,
                            uint32_t gap_cap, uint32_t *__restrict__ cs)
{
    uint32_t cnt = *gap_count;
    if (cnt > gap_cap) cnt = gap_cap;
    int64_t stride = (int64_t)gridDim.x * BLOCK;
    int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    for (uint32_t e = 0; e < cnt; ++e) {
        int64_t lo = gap_list[3 * e], hi = (int64_t)gap_list[3 * e + 1] + 1;
        uint32_t v = gap_list[3 * e + 2];
        for (int64_t c = lo + t; c < hi; c += stride) cs[c] = v;
    }
}

// ---- CIC deposit ---------------------------------------------------------------------------
//
// One thread per mesh cell.  Cell (i,j,k) receives mass from the particles of
// the eight cells (i-di, j-dj, k-dk), di,dj,dk in {0,1}, each through the
// corner (di,dj,dk) of its own cloud (functions.py:56-68).  With the particles
// in cell order those are eight contiguous runs, found through cell_start.  The
// sum is taken in float64 in a fixed order (corner order of functions.py:61-68,
// particles in cell order inside a corner) and rounded to float32 once, so the
// result is deterministic and no atomics are used.  A run longer than RUN_LANE
// is summed by the whole wave (stride-64 partial sums, then a fixed butterfly).
constexpr uint32_t RUN_LANE = 32;

template <typename TP>
__device__ __forceinline__ double corner_mass(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z,
                                              const float *__restrict__ m, uint32_t r, int di, int dj, int dk,
                                              const pm_geom &g)
{
    cell_t c = locate((double)x[r], (double)y[r], (double)z[r], g);
    double wx = di ? c.dx : 1.0 - c.dx;
    double wy = dj ? c.dy : 1.0 - c.dy;
    double wz = dk ? c.dz : 1.0 - c.dz;
    return (double)m[r] * wx * wy * wz;      // mass * wx * wy * wz, left to right (functions.py:61)
}

template <typename TP>
__global__ void __launch_bounds__(BLOCK)
k_deposit(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z, const float *__restrict__ m,
          const uint32_t *__restrict__ cs, float *__restrict__ density, pm_geom g)
{
    int64_t cell = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    bool valid = cell < g.M;
    int G = g.G;
    int k = (int)(cell % G), j = (int)((cell / G) % G), i = (int)(cell / ((int64_t)G * G));
    double acc = 0.0;
    // corner order of functions.py:61-68: 000 100 010 001 110 101 011 111
    constexpr int DI[8] = {0, 1, 0, 0, 1, 1, 0, 1}, DJ[8] = {0, 0, 1, 0, 1, 0, 1, 1}, DK[8] = {0, 0, 0, 1, 0, 1, 1, 1};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int di = DI[q], dj = DJ[q], dk = DK[q];
        uint32_t beg = 0, end = 0;
        if (valid) {
            int si = i - di, sj = j - dj, sk = k - dk;
            if (si < 0) si += G;
            if (sj < 0) sj += G;
            if (sk < 0) sk += G;
            int64_t s = ((int64_t)si * G + sj) * G + sk;
            beg = cs[s]; end = cs[s + 1];
        }
        uint32_t cnt = end - beg;
        if (cnt <= RUN_LANE)
            for (uint32_t r = beg; r < end; ++r) acc += corner_mass(x, y, z, m, r, di, dj, dk, g);
        unsigned long long big = __ballot(cnt > RUN_LANE);
        while (big) {
            int l = __ffsll((long long)big) - 1;
            big &= big - 1;
            uint32_t b = __shfl(beg, l), e = __shfl(end, l);
            double part = 0.0;
            for (uint32_t r = b + lane_id(); r < e; r += 64) part += corner_mass(x, y, z, m, r, di, dj, dk, g);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (lane_id() == l) acc += part;
        }
    }
    if (valid) density[cell] = (float)acc;
}

// ---- CIC gather + kick + drift + wrap ------------------------------------------------------

// np.gradient of the potential at node (i,j,k) along one axis (main.py:54):
// central difference times 0.5 inside, one-sided first difference on the faces.
__device__ __forceinline__ float diff1(const float *__restrict__ f, int64_t q, int64_t stride, int a, int G)
{
    if (a == 0) return f[q + stride] - f[q];
    if (a == G - 1) return f[q] - f[q - stride];
    return (f[q + stride] - f[q - stride]) * 0.5f;
}

template <typename TP, typename TV, bool FUSED>
__global__ void __launch_bounds__(BLOCK)
k_gather(const TP *__restrict__ px, const TP *__restrict__ py, const TP *__restrict__ pz,
         float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz,
         TV *__restrict__ vx, TV *__restrict__ vy, TV *__restrict__ vz,
         const float *__restrict__ gx, const float *__restrict__ gy, const float *__restrict__ gz,
         const float *__restrict__ phi, float *__restrict__ forces, int64_t n, pm_geom g, double dt, bool integrate)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const int G = g.G;
    double x = (double)px[r], y = (double)py[r], z = (double)pz[r];
    cell_t c = locate(x, y, z, g);
    int i1 = (c.i + 1 == G) ? 0 : c.i + 1, j1 = (c.j + 1 == G) ? 0 : c.j + 1, k1 = (c.k + 1 == G) ? 0 : c.k + 1;
    double wx[2] = {1.0 - c.dx, c.dx}, wy[2] = {1.0 - c.dy, c.dy}, wz[2] = {1.0 - c.dz, c.dz};
    int ii[2] = {c.i, i1}, jj[2] = {c.j, j1}, kk[2] = {c.k, k1};
    float fx = 0.0f, fy = 0.0f, fz = 0.0f;
    // corner order of functions.py:140-178; each add rounds the float32 accumulator
    constexpr int DI[8] = {0, 1, 0, 0, 1, 1, 0, 1}, DJ[8] = {0, 0, 1, 0, 1, 0, 1, 1}, DK[8] = {0, 0, 0, 1, 0, 1, 1, 1};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int a = ii[DI[q]], b = jj[DJ[q]], cc = kk[DK[q]];
        int64_t node = ((int64_t)a * G + b) * G + cc;
        double w = wx[DI[q]] * wy[DJ[q]] * wz[DK[q]];
        float ga, gb, gc;
        if (FUSED) {
            ga = diff1(phi, node, (int64_t)G * G, a, G);
            gb = diff1(phi, node, G, b, G);
            gc = diff1(phi, node, 1, cc, G);
        } else {
            ga = gx[node]; gb = gy[node]; gc = gz[node];
        }
        fx = (float)((double)fx + (double)ga * w);
        fy = (float)((double)fy + (double)gb * w);
        fz = (float)((double)fz + (double)gc * w);
    }
    if (forces) { forces[3 * r] = fx; forces[3 * r + 1] = fy; forces[3 * r + 2] = fz; }
    if (!integrate) return;
    // main.py:58-60
    const float dtf = (float)dt;
    float dvx = fx * dtf, dvy = fy * dtf, dvz = fz * dtf;      // float32 product (weak Python float)
    // velocities += forces*dt : the add runs in the velocities' dtype
    TV ux, uy, uz;
    if constexpr (sizeof(TV) == 4) { ux = vx[r] + dvx; uy = vy[r] + dvy; uz = vz[r] + dvz; }
    else { ux = vx[r] + (double)dvx; uy = vy[r] + (double)dvy; uz = vz[r] + (double)dvz; }
    vx[r] = ux; vy[r] = uy; vz[r] = uz;
    // positions += velocities*dt : float32 product for float32 velocities, the add in the wider
    // of the two dtypes, the result cast to float32 (on store, or by .astype at main.py:60)
    if constexpr (sizeof(TV) == 4 && sizeof(TP) == 4) {
        ox[r] = wrap_f32(px[r] + ux * dtf, g.Lf);
        oy[r] = wrap_f32(py[r] + uy * dtf, g.Lf);
        oz[r] = wrap_f32(pz[r] + uz * dtf, g.Lf);
    } else {
        double mx, my, mz;
        if constexpr (sizeof(TV) == 4) { mx = (double)(ux * dtf); my = (double)(uy * dtf); mz = (double)(uz * dtf); }
        else { mx = ux * dt; my = uy * dt; mz = uz * dt; }
        ox[r] = wrap_f32((float)(x + mx), g.Lf);
        oy[r] = wrap_f32((float)(y + my), g.Lf);
        oz[r] = wrap_f32((float)(z + mz), g.Lf);
    }
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

}  // namespace

// ---- launchers -----------------------------------------------------------------------------

hipError_t pm_launch_aos_to_soa(pm_ctx *c, const void *aos, bool f64, void *const dst[3], int64_t n)
{
    if (n == 0) return hipSuccess;
    if (f64)
        k_aos_to_soa<double, double><<<grid_for(n), BLOCK, 0, c->stream>>>((const double *)aos, (double *)dst[0], (double *)dst[1], (double *)dst[2], n);
    else
        k_aos_to_soa<float, float><<<grid_for(n), BLOCK, 0, c->stream>>>((const float *)aos, (float *)dst[0], (float *)dst[1], (float *)dst[2], n);
    return hipGetLastError();
}

hipError_t pm_launch_fill_particles_defaults(pm_ctx *c, bool zero_vel, bool ones_mass)
{
    int64_t n = c->n;
    if (n == 0) return hipSuccess;
    k_iota<<<grid_for(n), BLOCK, 0, c->stream>>>(c->id[c->cur], n);
    if (ones_mass) k_fill<float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->m[c->cur], 1.0f, n);
    if (zero_vel)
        for (int a = 0; a < 3; ++a) {
            if (c->vel_f64) k_fill<double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->vd[c->cur][a], 0.0, n);
            else k_fill<float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->vf[c->cur][a], 0.0f, n);
        }
    return hipGetLastError();
}

hipError_t pm_launch_cell_keys(pm_ctx *c)
{
    int64_t n = c->n;
    if (n == 0) return hipSuccess;
    int u = c->cur;
    if (c->pos_f64)
        k_cell_keys<double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->xd[u][0], c->xd[u][1], c->xd[u][2], c->key_in, n, c->g);
    else
        k_cell_keys<float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->x[u][0], c->x[u][1], c->x[u][2], c->key_in, n, c->g);
    return hipGetLastError();
}

hipError_t pm_launch_reorder(pm_ctx *c)
{
    int64_t n = c->n;
    if (n == 0) return hipSuccess;
    int u = c->cur, w = 1 - u;
    reorder_args a{};
    for (int d = 0; d < 3; ++d) {
        a.x_in[d] = c->x[u][d]; a.x_out[d] = c->x[w][d];
        a.xd_in[d] = c->xd[u][d]; a.xd_out[d] = c->xd[w][d];
        a.vf_in[d] = c->vf[u][d]; a.vf_out[d] = c->vf[w][d];
        a.vd_in[d] = c->vd[u][d]; a.vd_out[d] = c->vd[w][d];
    }
    a.m_in = c->m[u]; a.m_out = c->m[w];
    a.id_in = c->id[u]; a.id_out = c->id[w];
    a.src = c->src; a.n = n; a.pos_f64 = c->pos_f64; a.vel_f64 = c->vel_f64;
    k_reorder<<<grid_for(n), BLOCK, 0, c->stream>>>(a);
    c->cur = w;
    return hipGetLastError();
}

hipError_t pm_launch_cell_start(pm_ctx *c)
{
    hipError_t e = hipMemsetAsync(c->gap_count, 0, sizeof(uint32_t), c->stream);
    if (e != hipSuccess) return e;
    k_cell_start<<<grid_for(c->n + 1), BLOCK, 0, c->stream>>>(c->key_out, c->n, c->g.M, c->cell_start, c->gap_list, c->gap_count, c->gap_cap);
    k_fill_gaps<<<2048, BLOCK, 0, c->stream>>>(c->gap_list, c->gap_count, c->gap_cap, c->cell_start);
    return hipGetLastError();
}

hipError_t pm_launch_deposit(pm_ctx *c)
{
    int u = c->cur;
    if (c->pos_f64)
        k_deposit<double><<<grid_for(c->g.M), BLOCK, 0, c->stream>>>(c->xd[u][0], c->xd[u][1], c->xd[u][2], c->m[u], c->cell_start, c->density, c->g);
    else
        k_deposit<float><<<grid_for(c->g.M), BLOCK, 0, c->stream>>>(c->x[u][0], c->x[u][1], c->x[u][2], c->m[u], c->cell_start, c->density, c->g);
    return hipGetLastError();
}

template <typename TP, typename TV>
static void gather_dispatch(pm_ctx *c, const TP *const p[3], TV *const v[3], double dt, bool integrate, float *forces, bool fused)
{
    int u = c->cur;
    int64_t n = c->n;
    if (fused)
        k_gather<TP, TV, true><<<grid_for(n), BLOCK, 0, c->stream>>>(p[0], p[1], p[2], c->x[u][0], c->x[u][1], c->x[u][2], v[0], v[1], v[2],
                                                                     nullptr, nullptr, nullptr, c->potential, forces, n, c->g, dt, integrate);
    else
        k_gather<TP, TV, false><<<grid_for(n), BLOCK, 0, c->stream>>>(p[0], p[1], p[2], c->x[u][0], c->x[u][1], c->x[u][2], v[0], v[1], v[2],
                                                                      c->grad[0], c->grad[1], c->grad[2], nullptr, forces, n, c->g, dt, integrate);
}

hipError_t pm_launch_gather(pm_ctx *c, double dt, bool integrate, bool store_forces, bool fused_gradient)
{
    if (c->n == 0) return hipSuccess;
    int u = c->cur;
    float *f = store_forces ? c->forces : nullptr;
    if (c->pos_f64) {
        if (c->vel_f64) gather_dispatch<double, double>(c, c->xd[u], c->vd[u], dt, integrate, f, fused_gradient);
        else gather_dispatch<double, float>(c, c->xd[u], c->vf[u], dt, integrate, f, fused_gradient);
    } else {
        if (c->vel_f64) gather_dispatch<float, double>(c, c->x[u], c->vd[u], dt, integrate, f, fused_gradient);
        else gather_dispatch<float, float>(c, c->x[u], c->vf[u], dt, integrate, f, fused_gradient);
    }
    return hipGetLastError();
}

hipError_t pm_launch_export_particles(pm_ctx *c, void *pos_aos, bool pos_f64, void *vel_aos)
{
    int64_t n = c->n;
    if (n == 0) return hipSuccess;
    int u = c->cur;
    if (pos_aos) {
        if (pos_f64)
            k_soa_to_aos_by_id<double, double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->xd[u][0], c->xd[u][1], c->xd[u][2], c->id[u], (double *)pos_aos, n);
        else
            k_soa_to_aos_by_id<float, float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->x[u][0], c->x[u][1], c->x[u][2], c->id[u], (float *)pos_aos, n);
    }
    if (vel_aos) {
        if (c->vel_f64)
            k_soa_to_aos_by_id<double, double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->vd[u][0], c->vd[u][1], c->vd[u][2], c->id[u], (double *)vel_aos, n);
        else
            k_soa_to_aos_by_id<float, float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->vf[u][0], c->vf[u][1], c->vf[u][2], c->id[u], (float *)vel_aos, n);
    }
    return hipGetLastError();
}

hipError_t pm_launch_export_forces(pm_ctx *c, float *aos)
{
    if (c->n == 0) return hipSuccess;
    k_aos_by_id<<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->forces, c->id[c->cur], aos, c->n);
    return hipGetLastError();
}

hipError_t pm_launch_export_keys(pm_ctx *c, int64_t *keys, int64_t *perm)
{
    if (c->n == 0) return hipSuccess;
    k_export_keys<<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->key_out, c->id[c->cur], keys, perm, c->n);
    return hipGetLastError();
}
