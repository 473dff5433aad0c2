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

// row_start[row] = number of particles whose mesh row (key / G = i*G + j) is < row, for row in
// [0, G*G]: thread r (0..n) owns the rows in (row(key[r-1]), row(key[r])].  Short runs of empty
// rows are written by the owning lane, longer ones by its wave.
//
// The same pass finds the "heavy" cells (more than HEAVY_CELL particles): the thread of a
// cell's first particle checks key[r + HEAVY_CELL], finds the end of the cell's run by a
// galloping search, and lists the run in pieces of HEAVY_PIECE particles for k_heavy_cells.
// It also clears skip[] (k_heavy_cells then marks the particles of heavy cells in it).
constexpr uint32_t HEAVY_CELL = 64, HEAVY_PIECE = 2048;
constexpr uint32_t SKIP_HEAD = 0x80000000u;

__global__ void k_row_start(const uint32_t *__restrict__ key, int64_t n, uint32_t G, uint32_t R, uint32_t *__restrict__ rs,
                            uint32_t *__restrict__ skip, uint32_t *__restrict__ cell_list, uint32_t *__restrict__ cell_count)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    uint32_t lo = 0, hi = 0;
    if (r <= n) {
        uint32_t kprev = (r == 0) ? 0u : key[r - 1];
        uint32_t kcur = (r == n) ? 0u : key[r];
        lo = (r == 0) ? 0u : kprev / G + 1u;
        hi = (r == n) ? R + 1u : kcur / G + 1u;
        if (r < n) {
            skip[r] = 0u;
            bool head = (r == 0) || kprev != kcur;
            if (head && r + HEAVY_CELL < n && key[r + HEAVY_CELL] == kcur) {
                // first index past the run: gallop, then bisect; inside = last index known to hold kcur
                int64_t inside = r + HEAVY_CELL, step = HEAVY_CELL, outside = n;
                while (inside + step < n) {
                    if (key[inside + step] == kcur) { inside += step; step <<= 1; }
                    else { outside = inside + step; break; }
                }
                while (outside - inside > 1) {
                    int64_t mid = (inside + outside) >> 1;
                    if (key[mid] == kcur) inside = mid; else outside = mid;
                }
                uint32_t len = (uint32_t)(outside - r);
                uint32_t pieces = (len + HEAVY_PIECE - 1) / HEAVY_PIECE;
                uint32_t e = atomicAdd(cell_count, pieces);           // at most n / HEAVY_CELL entries in all
                for (uint32_t p = 0; p < pieces; ++p) {
                    cell_list[3 * (e + p)] = (uint32_t)r; cell_list[3 * (e + p) + 1] = (uint32_t)outside; cell_list[3 * (e + p) + 2] = p;
                }
            }
        }
    }
    uint32_t len = hi - lo;
    if (len > 0 && len <= 32u)
        for (uint32_t c = lo; c < hi; ++c) rs[c] = (uint32_t)r;
    unsigned long long big = __ballot(len > 32u);
    while (big) {
        int l = __ffsll((long long)big) - 1;
        big &= big - 1;
        uint32_t a = __shfl(lo, l), b = __shfl(hi, l), v = (uint32_t)__shfl(r, l);
        for (uint32_t c = a + lane_id(); c < b; c += 64) rs[c] = v;
    }
}

// ---- CIC deposit ---------------------------------------------------------------------------
//
// One wave per mesh row (i,j,:) of the density.  The row receives mass from the particles of
// the four mesh rows (i-di, j-dj), di,dj in {0,1}, through the corners (di,dj,0) into cell k and
// (di,dj,1) into cell k+1 of their clouds (functions.py:56-68).  With the particles in cell
// order each of those rows is one contiguous run of the particle arrays, found through
// row_start.  The wave walks the four runs, 64 particles at a time, and adds the float64 corner
// masses into a float64 copy of the row held in LDS; the row is rounded to float32 once, when
// it is written out.  No global atomics, no mesh-sized index: the only mesh traffic is the one
// coalesced write of the density.
//
// All adds to a row come from one wave, in program order, so the sum has one fixed order.
// Two things keep dense clumps from serialising the walk:
//  * heavy cells (more than HEAVY_CELL particles) are summed once, for all eight corners, by
//    k_heavy_cells (one wave per piece of the cell's run, fixed butterfly); the walk then takes
//    the cell's sums from pre[] at the cell's first particle and jumps over the rest;
//  * rows still fed by more than HEAVY_ROW particles are left to k_deposit_heavy, where the
//    waves of a block each walk a contiguous slice of the runs into their own LDS row and the
//    slices are added in wave order.
constexpr uint32_t HEAVY_ROW = 2048;

// The eight corner masses of particle r: mass * wx * wy * wz, left to right (functions.py:61-68),
// indexed q = di + 2*dj + 4*dk.
template <typename TP>
__device__ __forceinline__ void corner_masses(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z,
                                              const float *__restrict__ m, uint32_t r, const pm_geom &g, double w[8])
{
    cell_t c = locate((double)x[r], (double)y[r], (double)z[r], g);
    double mm = (double)m[r];
    double mx[2] = {mm * (1.0 - c.dx), mm * c.dx};
    double wy[2] = {1.0 - c.dy, c.dy}, wz[2] = {1.0 - c.dz, c.dz};
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = mx[q & 1] * wy[(q >> 1) & 1] * wz[q >> 2];
}

// One wave per listed piece of a heavy cell's run: the eight corner sums of the piece go to
// pre[q*cap + first particle of the cell + piece number]; skip[] gets, for every particle of the
// piece, the index one past the cell's run (the cell's first particle also gets SKIP_HEAD).
template <typename TP>
__global__ void __launch_bounds__(BLOCK)
k_heavy_cells(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z, const float *__restrict__ m,
              const uint32_t *__restrict__ cell_list, const uint32_t *__restrict__ cell_count,
              double *__restrict__ pre, uint32_t *__restrict__ skip, int64_t cap, pm_geom g)
{
    const uint32_t cnt = *cell_count;
    const uint32_t waves = gridDim.x * (BLOCK / 64);
    for (uint32_t e = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); e < cnt; e += waves) {
        const uint32_t beg = cell_list[3 * e], end = cell_list[3 * e + 1], piece = cell_list[3 * e + 2];
        const uint32_t pb = beg + piece * HEAVY_PIECE;
        const uint32_t pe = (end - pb > HEAVY_PIECE) ? pb + HEAVY_PIECE : end;
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t r = pb + lane_id(); r < pe; r += 64) {
            double w[8];
            corner_masses(x, y, z, m, r, g, w);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += w[q];
            skip[r] = (r == beg) ? (end | SKIP_HEAD) : end;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_xor(acc[q], off);
        }
        if (lane_id() < 8) {
            double v = acc[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) v = (lane_id() == q) ? acc[q] : v;
            pre[(int64_t)lane_id() * cap + beg + piece] = v;
        }
    }
}

// Mass of particle r entering cells k (lo) and k+1 (hi) of the row displaced by (di,dj):
// mass * wx * wy * wz, left to right (functions.py:61-68).
template <typename TP>
__device__ __forceinline__ void row_masses(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z,
                                           const float *__restrict__ m, uint32_t r, int di, int dj, const pm_geom &g,
                                           int &k, double &lo, double &hi)
{
    cell_t c = locate((double)x[r], (double)y[r], (double)z[r], g);
    double w = (double)m[r] * (di ? c.dx : 1.0 - c.dx) * (dj ? c.dy : 1.0 - c.dy);
    k = c.k;
    lo = w * (1.0 - c.dz);
    hi = w * c.dz;
}

__device__ __forceinline__ void lds_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One run of a source row into the LDS row `acc`, 64 particles per pass.  The particles are in
// cell order, so the lanes of a pass hold non-decreasing k and the lanes that share a cell are
// neighbours.  Where many lanes share cells their masses are first summed by a segmented scan
// over the wave (fixed tree order) and only the last lane of every cell adds to LDS; otherwise
// every lane adds its own.  Particles of a heavy cell are represented by the cell's first
// particle alone, which adds the sums k_heavy_cells left in pre[]; when a pass ends inside a
// heavy cell the walk continues after that cell.
template <typename TP>
__device__ __forceinline__ void deposit_run(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z,
                                            const float *__restrict__ m, const uint32_t *__restrict__ key,
                                            const uint32_t *__restrict__ skip, const double *__restrict__ pre, int64_t cap,
                                            uint32_t beg, uint32_t end, int di, int dj, const pm_geom &g, double *acc)
{
    const int G = g.G;
    const int lane = lane_id();
    const double *pre_lo = pre + (int64_t)(di + 2 * dj) * cap, *pre_hi = pre + (int64_t)(di + 2 * dj + 4) * cap;
    uint32_t base = beg;
    while (base < end) {                                        // wave-uniform
        const uint32_t r = base + lane;
        const bool live = r < end;
        const uint32_t sk = live ? skip[r] : 0u;
        int k = 0x7fffffff - lane;                              // idle lanes: distinct, never equal to a cell
        double lo = 0.0, hi = 0.0;
        bool add = false;
        if (live && sk == 0u) {
            row_masses(x, y, z, m, r, di, dj, g, k, lo, hi);
            add = true;
        } else if (sk & SKIP_HEAD) {
            const uint32_t cend = sk & ~SKIP_HEAD;
            k = (int)(key[r] % (uint32_t)G);
            for (uint32_t p = r; p < r + (cend - r + HEAVY_PIECE - 1) / HEAVY_PIECE; ++p) { lo += pre_lo[p]; hi += pre_hi[p]; }
            add = true;
        }
        int kprev = __shfl_up(k, 1);
        unsigned long long dup = __ballot(lane > 0 && kprev == k);
        if (__popcll(dup) > 8) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                int ku = __shfl_up(k, d);
                double lu = __shfl_up(lo, d), hu = __shfl_up(hi, d);
                if (lane >= d && ku == k) { lo += lu; hi += hu; }
            }
            int knext = __shfl_down(k, 1);
            add = add && (lane == 63 || knext != k);             // last lane of its cell: holds the cell's sum
        }
        if (add) {
            lds_add(&acc[k], lo);
            lds_add(&acc[(k + 1 == G) ? 0 : k + 1], hi);
        }
        const uint32_t sk_last = __shfl(sk, 63) & ~SKIP_HEAD;   // 0 unless the pass ended inside a heavy cell
        base = (sk_last != 0u) ? sk_last : base + 64;
    }
}

struct row_runs { uint32_t beg[4], end[4], total; };

// The four source rows of density row `row`, in the order (di,dj) = (0,0) (1,0) (0,1) (1,1).
__device__ __forceinline__ row_runs runs_of(const uint32_t *__restrict__ rs, uint32_t row, int G)
{
    int i = (int)(row / (uint32_t)G), j = (int)(row % (uint32_t)G);
    int im = (i == 0) ? G - 1 : i - 1, jm = (j == 0) ? G - 1 : j - 1;
    uint32_t src[4] = {(uint32_t)(i * G + j), (uint32_t)(im * G + j), (uint32_t)(i * G + jm), (uint32_t)(im * G + jm)};
    row_runs t;
    t.total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t.beg[q] = rs[src[q]]; t.end[q] = rs[src[q] + 1];
        t.total += t.end[q] - t.beg[q];
    }
    return t;
}

template <typename TP>
__global__ void __launch_bounds__(BLOCK)
k_deposit_rows(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z, const float *__restrict__ m,
               const uint32_t *__restrict__ key, const uint32_t *__restrict__ skip, const double *__restrict__ pre, int64_t cap,
               const uint32_t *__restrict__ rs, float *__restrict__ density, uint32_t *__restrict__ heavy_rows,
               uint32_t *__restrict__ heavy_count, pm_geom g)
{
    extern __shared__ double lds_rows[];
    const int G = g.G;
    const uint32_t R = (uint32_t)G * (uint32_t)G;
    const uint32_t row = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (row >= R) return;                              // whole waves leave together; no barrier below
    double *acc = lds_rows + (size_t)(threadIdx.x >> 6) * G;
    float *out = density + (size_t)row * G;
    row_runs t = runs_of(rs, row, G);
    if (t.total > HEAVY_ROW) {
        if (lane_id() == 0) heavy_rows[atomicAdd(heavy_count, 1u)] = row;
        return;
    }
    if (t.total == 0) {
        for (int c = lane_id(); c < G; c += 64) out[c] = 0.0f;
        return;
    }
    for (int c = lane_id(); c < G; c += 64) acc[c] = 0.0;
    deposit_run(x, y, z, m, key, skip, pre, cap, t.beg[0], t.end[0], 0, 0, g, acc);
    deposit_run(x, y, z, m, key, skip, pre, cap, t.beg[1], t.end[1], 1, 0, g, acc);
    deposit_run(x, y, z, m, key, skip, pre, cap, t.beg[2], t.end[2], 0, 1, g, acc);
    deposit_run(x, y, z, m, key, skip, pre, cap, t.beg[3], t.end[3], 1, 1, g, acc);
    for (int c = lane_id(); c < G; c += 64) out[c] = (float)acc[c];
}

template <typename TP>
__global__ void __launch_bounds__(1024)
k_deposit_heavy(const TP *__restrict__ x, const TP *__restrict__ y, const TP *__restrict__ z, const float *__restrict__ m,
                const uint32_t *__restrict__ key, const uint32_t *__restrict__ skip, const double *__restrict__ pre, int64_t cap,
                const uint32_t *__restrict__ rs, float *__restrict__ density, const uint32_t *__restrict__ heavy_rows,
                const uint32_t *__restrict__ heavy_count, pm_geom g)
{
    extern __shared__ double lds_rows[];
    const int G = g.G;
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6;
    double *acc = lds_rows + (size_t)w * G;
    const uint32_t cnt = *heavy_count;
    for (uint32_t e = blockIdx.x; e < cnt; e += gridDim.x) {
        const uint32_t row = heavy_rows[e];
        row_runs t = runs_of(rs, row, G);
        for (int c = lane_id(); c < G; c += 64) acc[c] = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // wave w takes the w-th slice of the run; slices are whole multiples of 64 particles
            uint32_t len = t.end[q] - t.beg[q];
            uint32_t sl = ((len + nw - 1) / nw + 63u) & ~63u;
            uint32_t b = t.beg[q] + (uint32_t)w * sl, e2 = b + sl;
            if (b > t.end[q]) b = t.end[q];
            if (e2 > t.end[q]) e2 = t.end[q];
            deposit_run(x, y, z, m, key, skip, pre, cap, b, e2, q & 1, q >> 1, g, acc);
        }
        __syncthreads();
        float *out = density + (size_t)row * G;
        for (int c = threadIdx.x; c < G; c += blockDim.x) {
            double sum = 0.0;
            for (int v = 0; v < nw; ++v) sum += lds_rows[(size_t)v * G + c];
            out[c] = (float)sum;
        }
        __syncthreads();
    }
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

hipError_t pm_launch_row_start(pm_ctx *c)
{
    const uint32_t G = (uint32_t)c->g.G;
    hipError_t e = hipMemsetAsync(c->heavy_count, 0, 2 * sizeof(uint32_t), c->stream);     // [0] heavy rows, [1] heavy-cell pieces
    if (e != hipSuccess) return e;
    k_row_start<<<grid_for(c->n + 1), BLOCK, 0, c->stream>>>(c->key_out, c->n, G, G * G, c->row_start, c->skip, c->cell_list, c->heavy_count + 1);
    return hipGetLastError();
}

int pm_heavy_waves(int G)
{
    int nw = (int)((128u * 1024u) / ((size_t)G * sizeof(double)));
    return nw > 16 ? 16 : (nw < 1 ? 1 : nw);
}

template <typename TP>
static hipError_t deposit_dispatch(pm_ctx *c, const TP *const p[3])
{
    const int G = c->g.G;
    const uint32_t R = (uint32_t)G * (uint32_t)G;
    const int u = c->cur;
    hipError_t e = hipMemsetAsync(c->heavy_count, 0, sizeof(uint32_t), c->stream);
    if (e != hipSuccess) return e;
    k_heavy_cells<TP><<<1024, BLOCK, 0, c->stream>>>(p[0], p[1], p[2], c->m[u], c->cell_list, c->heavy_count + 1, c->pre, c->skip, c->cap, c->g);
    const size_t lds_light = (size_t)(BLOCK / 64) * G * sizeof(double);
    k_deposit_rows<TP><<<dim3((R + BLOCK / 64 - 1) / (BLOCK / 64)), BLOCK, lds_light, c->stream>>>(
        p[0], p[1], p[2], c->m[u], c->key_out, c->skip, c->pre, c->cap, c->row_start, c->density, c->heavy_rows, c->heavy_count, c->g);
    const int nw = pm_heavy_waves(G);
    const size_t lds_heavy = (size_t)nw * G * sizeof(double);
    static bool attr_set[2] = {false, false};
    if (!attr_set[sizeof(TP) == 8]) {
        e = hipFuncSetAttribute((const void *)k_deposit_heavy<TP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[sizeof(TP) == 8] = true;
    }
    k_deposit_heavy<TP><<<512, nw * 64, lds_heavy, c->stream>>>(p[0], p[1], p[2], c->m[u], c->key_out, c->skip, c->pre, c->cap,
                                                                c->row_start, c->density, c->heavy_rows, c->heavy_count, c->g);
    return hipGetLastError();
}

hipError_t pm_launch_deposit(pm_ctx *c)
{
    int u = c->cur;
    if (c->pos_f64) return deposit_dispatch<double>(c, c->xd[u]);
    return deposit_dispatch<float>(c, c->x[u]);
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
