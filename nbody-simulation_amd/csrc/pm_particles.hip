// Particle-side kernels of the PM step for gfx950: cell keys, reorder into cell order, row-start
// table, CIC deposit (row-centric, LDS accumulation, no global atomics), CIC gather fused with
// kick / drift / wrap, and the host-copy helpers.
//
// Arithmetic follows what numba compiles the reference loops to (functions.py:19-70, 99-180):
// wrap, scale, fractional parts and weights in float64; this file is built with
// -ffp-contract=off so no product is fused into an add.
#include "pm_internal.hpp"
#include "pm_fft_dev.hpp"

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

// functions.py:27-43.  A scaled coordinate that rounds up to G (the reference would index out
// of bounds there) is folded back to cell 0 of that axis.
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

template <typename TP>
__device__ __forceinline__ cell_t locate(const pm_prec<TP> &p, const pm_geom &g)
{
    return locate((double)p.x, (double)p.y, (double)p.z, g);
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

// ---- host copies ---------------------------------------------------------------------------

template <typename TP>
__global__ void k_pack_pos(const TP *__restrict__ aos, const float *__restrict__ mass, pm_prec<TP> *__restrict__ P, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n) return;
    pm_prec<TP> rec{};
    rec.x = aos[3 * p]; rec.y = aos[3 * p + 1]; rec.z = aos[3 * p + 2];
    rec.m = mass ? mass[p] : 1.0f;
    P[p] = rec;
}

template <typename TV>
__global__ void k_pack_vel(const TV *__restrict__ aos, pm_vrec<TV> *__restrict__ V, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n) return;
    pm_vrec<TV> rec{};
    if (aos) { rec.x = aos[3 * p]; rec.y = aos[3 * p + 1]; rec.z = aos[3 * p + 2]; }
    rec.id = (uint32_t)p;
    V[p] = rec;
}

template <typename TP, typename TV>
__global__ void k_export_pos(const pm_prec<TP> *__restrict__ P, const pm_vrec<TV> *__restrict__ V, const uint32_t *__restrict__ vmap,
                             TP *__restrict__ aos, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    pm_prec<TP> rec = P[r];
    int64_t o = 3 * (int64_t)V[vmap ? vmap[r] : r].id;
    aos[o] = rec.x; aos[o + 1] = rec.y; aos[o + 2] = rec.z;
}

template <typename TV>
__global__ void k_export_vel(const pm_vrec<TV> *__restrict__ V, TV *__restrict__ aos, int64_t n)
{
    // velocity records carry their own particle index, so their storage order does not matter here
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    pm_vrec<TV> rec = V[r];
    int64_t o = 3 * (int64_t)rec.id;
    aos[o] = rec.x; aos[o + 1] = rec.y; aos[o + 2] = rec.z;
}

template <typename TV>
__global__ void k_export_forces(const float *__restrict__ in, const pm_vrec<TV> *__restrict__ V, const uint32_t *__restrict__ vmap,
                                float *__restrict__ out, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    int64_t o = 3 * (int64_t)V[vmap ? vmap[r] : r].id;
    out[o] = in[3 * r]; out[o + 1] = in[3 * r + 1]; out[o + 2] = in[3 * r + 2];
}

template <typename TV>
__global__ void k_export_keys(const uint32_t *__restrict__ key, const pm_vrec<TV> *__restrict__ V, const uint32_t *__restrict__ vmap,
                              int64_t *__restrict__ keys, int64_t *__restrict__ perm, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    keys[r] = (int64_t)key[r];
    perm[r] = (int64_t)V[vmap ? vmap[r] : r].id;
}

// ---- binning -------------------------------------------------------------------------------

template <typename TP>
__global__ void k_cell_keys(const pm_prec<TP> *__restrict__ P, uint32_t *__restrict__ key, int64_t n, pm_geom g)
{
    int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= n) return;
    cell_t c = locate(P[p], g);
    key[p] = (uint32_t)(((int64_t)c.i * g.G + c.j) * g.G + c.k);
}

// Cell order -> storage order.  Slot r of the new storage takes the position record of old slot
// src[r] and gets a fraction record {dx, dy, dz, mass} -- the particle's offsets inside its cell
// (functions.py:41-43, computed in float64 and rounded to float32 once) -- which is all the deposit
// needs besides the cell key, so that the wrap / scale / floor arithmetic runs once per particle
// here and not once per visit there.  The velocity records are not moved here: the gather, which reads and writes every one of
// them anyway, reads record src[r] and writes it to slot r (k_move_v does it when no gather follows).
//
// The same pass prepares the deposit's handling of "heavy" cells (more than HEAVY_CELL
// particles): it clears skip[], flags the mesh rows that hold a heavy cell and cuts the runs of heavy
// cells into pieces for k_heavy_pieces.  A piece starts at the first particle of a heavy cell, and at
// every multiple of HEAVY_PIECE that has at least HEAVY_CELL particles of the same cell on either
// side (key[] is the sorted key array, so slot r reads its neighbours' keys).
#ifndef PM_HEAVY_CELL
#define PM_HEAVY_CELL 128
#endif
#ifndef PM_HEAVY_WAVES
#define PM_HEAVY_WAVES 8
#endif
#ifndef PM_HEAVY_RUN
#define PM_HEAVY_RUN 1024
#endif
#ifndef PM_HEAVY_GRID
#define PM_HEAVY_GRID 2048
#endif
constexpr uint32_t HEAVY_CELL = PM_HEAVY_CELL, HEAVY_PIECE = 2048;
// Lists filled through atomic counters are split into shards, each with its own counter on its own
// 128-byte line and its own region of the list, so that appends do not queue on one address.
constexpr uint32_t PIECE_SHARDS = 64, ROW_SHARDS = 16, COUNTER_STRIDE = 32;
// pieces a block of BLOCK consecutive particles can start: one per HEAVY_CELL + 1 particles, plus a cut
constexpr uint32_t PIECES_PER_BLOCK = BLOCK / (HEAVY_CELL + 1) + 2;
// skip[r]: 0 for a particle that adds its own two corner masses; for a particle of a piece, the index
// one past the piece, with SKIP_HEAD set on the piece's first particle (which adds the piece's sums)
constexpr uint32_t SKIP_HEAD = 0x80000000u;

// Is index c (a multiple of HEAVY_PIECE, inside a run of key k) the start of a new piece?
__device__ __forceinline__ bool piece_cut(const uint32_t *__restrict__ key, int64_t c, uint32_t k, int64_t n)
{
    return c >= HEAVY_CELL && c + HEAVY_CELL - 1 < n && key[c - HEAVY_CELL] == k && key[c + HEAVY_CELL - 1] == k;
}

template <typename TV>
__global__ void k_move_v(const pm_vrec<TV> *__restrict__ Vin, pm_vrec<TV> *__restrict__ Vout, const uint32_t *__restrict__ src, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r < n) Vout[r] = Vin[src[r]];
}

template <typename TP>
__global__ void k_reorder(const pm_prec<TP> *__restrict__ Pin, pm_prec<TP> *__restrict__ Pout,
                          pm_frac *__restrict__ F, const uint32_t *__restrict__ src, const uint32_t *__restrict__ key,
                          uint32_t *__restrict__ skip, uint8_t *__restrict__ row_flag, uint32_t *__restrict__ piece_list,
                          uint32_t *__restrict__ piece_count, uint32_t shard_cap, int64_t n, pm_geom g)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    bool piece = false;
    if (r < n) {
        int64_t s = src[r];
        const pm_prec<TP> rec = Pin[s];
        Pout[r] = rec;
        const cell_t c = locate(rec, g);
        F[r] = pm_frac{(float)c.dx, (float)c.dy, (float)c.dz, rec.m};
        skip[r] = 0u;
        uint32_t kcur = key[r];
        bool head = (r == 0) || key[r - 1] != kcur;
        if (head) {
            piece = (r + HEAVY_CELL < n) && key[r + HEAVY_CELL] == kcur;      // run longer than HEAVY_CELL
            if (piece) row_flag[kcur / (uint32_t)g.G] = 1;
        } else if (r % HEAVY_PIECE == 0) {
            piece = piece_cut(key, r, kcur, n);
        }
    }
    // one atomic per wave, on the counter of this block's shard (a shard holds at most
    // PIECES_PER_BLOCK entries per block that feeds it: shard_cap)
    unsigned long long pm = __ballot(piece);
    if (pm) {
        const uint32_t shard = blockIdx.x % PIECE_SHARDS;
        int leader = __ffsll((long long)pm) - 1;
        uint32_t base = 0;
        if (lane_id() == leader) base = atomicAdd(piece_count + shard * COUNTER_STRIDE, (uint32_t)__popcll(pm));
        base = __shfl(base, leader);
        if (piece) piece_list[(size_t)shard * shard_cap + base + __popcll(pm & ((1ull << lane_id()) - 1ull))] = (uint32_t)r;
    }
}

// row_start[row] = number of particles whose mesh row (key / G = i*G + j) is < row, for row in
// [0, G*G]: one bisection of the sorted keys per row.
__global__ void k_row_start(const uint32_t *__restrict__ key, int64_t n, uint32_t G, uint32_t R, uint32_t *__restrict__ rs)
{
    uint32_t row = blockIdx.x * BLOCK + threadIdx.x;
    if (row > R) return;
    const uint64_t target = (uint64_t)row * G;          // first key of the row
    int64_t lo = 0, hi = n;                             // first index in [lo, hi] with key >= target
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)key[mid] < target) lo = mid + 1; else hi = mid;
    }
    rs[row] = (uint32_t)lo;
}

// ---- CIC deposit ---------------------------------------------------------------------------
//
// One wave per mesh row (i,j,:) of the density.  The row receives mass from the particles of
// the four mesh rows (i-di, j-dj), di,dj in {0,1}, through the corners (di,dj,0) into cell k and
// (di,dj,1) into cell k+1 of their clouds (functions.py:56-68).  With the particles in cell
// order each of those rows is one contiguous run of the particle array, found through
// row_start.  A wave walks the four runs as one sequence, 64 particles at a time, and adds the
// float64 corner masses into a float64 copy of the row held in LDS; the row is rounded to float32
// once, when it is written out.  No global atomics, no mesh-sized index: the only mesh traffic is
// the one coalesced write of the density.
//
// All adds to an LDS row come from one wave, in program order, and the LDS rows are added in a
// fixed order, so the sum has one fixed order.
// Two things keep dense clumps from serialising the walk:
//  * the runs of heavy cells (more than HEAVY_CELL particles) are summed once, for all eight
//    corners, by k_heavy_pieces (one wave per piece, fixed butterfly); a walk takes a piece's sums
//    from pre[] at the piece's first particle and does nothing for its other particles;
//  * rows with a source run of more than HEAVY_RUN particles, or with a heavy cell in a source row,
//    are left to k_deposit_heavy, where several waves walk contiguous slices of each run into their
//    own LDS rows and the rows are added in wave order.
constexpr uint32_t HEAVY_RUN = PM_HEAVY_RUN;

// The eight corner masses of a fraction record, mass * wx * wy * wz left to right (functions.py:61-68),
// q = di + 2*dj + 4*dk; weights in float64 from the float32 offsets.
__device__ __forceinline__ void corner_masses(const pm_frac &f, double w[8])
{
    const double dx = (double)f.dx, dy = (double)f.dy, dz = (double)f.dz, mm = (double)f.m;
    const double mx[2] = {mm * (1.0 - dx), mm * dx};
    const double wy[2] = {1.0 - dy, dy}, wz[2] = {1.0 - dz, dz};
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = mx[q & 1] * wy[(q >> 1) & 1] * wz[q >> 2];
}

// One wave per listed piece: the part of the run of its first particle's cell up to the next
// piece start.  Its eight corner sums go to pre[q*cap + first]; skip[] gets, for every particle of
// the piece, the index one past the piece (the first one also SKIP_HEAD).  Each lane sums every 64th
// particle, four loads in flight.
__global__ void __launch_bounds__(BLOCK)
k_heavy_pieces(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key, const uint32_t *__restrict__ piece_list,
               const uint32_t *__restrict__ piece_count, uint32_t shard_cap, double *__restrict__ pre,
               uint32_t *__restrict__ skip, int64_t n, int64_t cap)
{
    // blocks s, s + PIECE_SHARDS, ... serve shard s
    const uint32_t shard = blockIdx.x % PIECE_SHARDS;
    const uint32_t cnt = piece_count[shard * COUNTER_STRIDE];
    const uint32_t waves = (gridDim.x / PIECE_SHARDS) * (BLOCK / 64);
    piece_list += (size_t)shard * shard_cap;
    for (uint32_t e = (blockIdx.x / PIECE_SHARDS) * (BLOCK / 64) + (threadIdx.x >> 6); e < cnt; e += waves) {
        const uint32_t first = piece_list[e];
        const uint32_t k0 = key[first];
        // the piece ends where the run ends or at the next cut that starts a piece
        int64_t bound = n;
        for (int64_t c = ((int64_t)(first / HEAVY_PIECE) + 1) * HEAVY_PIECE; c < n; c += HEAVY_PIECE)
            if (key[c] != k0 || piece_cut(key, c, k0, n)) { bound = c; break; }
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t mine = 0;
        bool more = true;
        for (int64_t r0 = (int64_t)first + lane_id(); more; r0 += 256) {       // wave-uniform exit
            uint32_t kk[4];
            pm_frac ff[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t r = r0 + 64 * u;
                kk[u] = ~k0; ff[u] = pm_frac{0.f, 0.f, 0.f, 0.f};
                if (r < bound) { kk[u] = key[r]; ff[u] = F[r]; }
            }
            bool all = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (kk[u] == k0) {                           // cell order: the first mismatch ends the run
                    double w[8];
                    corner_masses(ff[u], w);
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] += w[q];
                    ++mine;
                } else {
                    all = false;
                }
            }
            more = __all(all);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        const uint32_t end = first + mine;
        for (uint32_t r = first + lane_id(); r < end; r += 64) skip[r] = (r == first) ? (end | SKIP_HEAD) : end;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_xor(acc[q], off);
        }
        if (lane_id() < 8) {
            double v = acc[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) v = (lane_id() == q) ? acc[q] : v;
            pre[(int64_t)lane_id() * cap + first] = v;
        }
    }
}

__device__ __forceinline__ void lds_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The lanes of a pass hold (k, lo, hi): cell and the masses for cells k and k+1 (idle lanes: a k no
// cell has, distinct per lane).  Lanes that share a cell are mostly neighbours (cell order).  Where
// many neighbours share cells, each stretch of equal k is first summed by a segmented scan over the
// wave (fixed tree order) and only its last lane adds to LDS; otherwise every lane adds its own.
__device__ __forceinline__ void add_pass(double *acc, int G, int k, double lo, double hi, bool add)
{
    const int lane = lane_id();
    int kprev = __shfl_up(k, 1);
    const unsigned long long dup = __ballot(lane > 0 && kprev == k);
    if (__popcll(dup) > 8) {
        const unsigned long long starts = ~dup;                 // lanes that begin a stretch (bit 0 is set)
        const int start = 63 - __clzll((long long)(starts & ((2ull << lane) - 1ull)));
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            double lu = __shfl_up(lo, d), hu = __shfl_up(hi, d);
            if (lane - d >= start) { lo += lu; hi += hu; }
        }
        add = add && (lane == 63 || ((starts >> (lane + 1)) & 1ull));   // last lane of its stretch: holds the sum
    }
    if (add) {
        lds_add(&acc[k], lo);
        lds_add(&acc[(k + 1 == G) ? 0 : k + 1], hi);
    }
}

struct row_runs { uint32_t beg[4], end[4], longest; bool flagged; };

// The four source rows of density row `row`, in the order (di,dj) = (0,0) (1,0) (0,1) (1,1).
__device__ __forceinline__ row_runs runs_of(const uint32_t *__restrict__ rs, const uint8_t *__restrict__ row_flag, uint32_t row, int G)
{
    int i = (int)(row / (uint32_t)G), j = (int)(row % (uint32_t)G);
    int im = (i == 0) ? G - 1 : i - 1, jm = (j == 0) ? G - 1 : j - 1;
    uint32_t src[4] = {(uint32_t)(i * G + j), (uint32_t)(im * G + j), (uint32_t)(i * G + jm), (uint32_t)(im * G + jm)};
    row_runs t;
    t.longest = 0;
    uint32_t fl = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t.beg[q] = rs[src[q]]; t.end[q] = rs[src[q] + 1];
        fl |= row_flag[src[q]];
        uint32_t len = t.end[q] - t.beg[q];
        t.longest = len > t.longest ? len : t.longest;
    }
    t.flagged = fl != 0;
    return t;
}

// What one lane holds of a 64-particle pass of a light row's walk.
struct light_t { uint32_t key; pm_frac f; };

// The four runs of a light row (no heavy cell) walked as one sequence of `total` particles: entry v
// is particle slot(v) of run q(v).
struct packed_runs {
    uint32_t o1, o2, o3, total, b0, b1, b2, b3;
    __device__ __forceinline__ explicit packed_runs(const row_runs &t)
    {
        o1 = t.end[0] - t.beg[0]; o2 = o1 + (t.end[1] - t.beg[1]); o3 = o2 + (t.end[2] - t.beg[2]);
        total = o3 + (t.end[3] - t.beg[3]);
        b0 = t.beg[0]; b1 = t.beg[1] - o1; b2 = t.beg[2] - o2; b3 = t.beg[3] - o3;     // modulo 2^32
    }
    __device__ __forceinline__ int run(uint32_t v) const { return (int)(v >= o1) + (int)(v >= o2) + (int)(v >= o3); }
    __device__ __forceinline__ uint32_t slot(uint32_t v, int q) const { return v + (q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3); }
    __device__ __forceinline__ light_t load(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key, uint32_t base) const
    {
        light_t t{};
        const uint32_t v = base + lane_id();
        if (v < total) { const uint32_t r = slot(v, run(v)); t.key = key[r]; t.f = F[r]; }
        return t;
    }
};

__device__ __forceinline__ void deposit_light(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key, const packed_runs &p,
                                              light_t cur, int G, double *acc)
{
    const int lane = lane_id();
    for (uint32_t base = 0; base < p.total; base += 64) {       // wave-uniform
        const uint32_t v = base + lane;
        const pm_frac f = cur.f;
        const int kc = (int)(cur.key % (uint32_t)G);
        if (base + 64 < p.total) cur = p.load(F, key, base + 64);   // in flight while this pass is summed
        int k = 0x7fffffff - lane;
        double lo = 0.0, hi = 0.0;
        const bool live = v < p.total;
        if (live) {
            const int q = p.run(v);
            const double dx = (double)f.dx, dy = (double)f.dy, dz = (double)f.dz;
            const double w = (double)f.m * ((q & 1) ? dx : 1.0 - dx) * ((q & 2) ? dy : 1.0 - dy);
            k = kc; lo = w * (1.0 - dz); hi = w * dz;
        }
        add_pass(acc, G, k, lo, hi, live);
    }
}

// What one lane holds of a 64-particle pass over a run of a heavy row.
struct heavy_t { uint32_t sk, key; pm_frac f; };

__device__ __forceinline__ heavy_t load_heavy(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key,
                                              const uint32_t *__restrict__ skip, uint32_t base, uint32_t end)
{
    heavy_t t{};
    t.sk = 1u;                                          // past the end: neither adds nor heads a piece
    const uint32_t r = base + lane_id();
    if (r < end) { t.sk = skip[r]; t.key = key[r]; t.f = F[r]; }
    return t;
}

constexpr int HEAD_LIST = 128;      // entries of a wave's list of piece heads (flushed at 64; a pass adds at most 64)

// Adds the sums of the listed pieces: entry = (first particle, cell k).
__device__ __forceinline__ void flush_heads(const uint2 *list, uint32_t cnt, const double *__restrict__ pre_lo,
                                            const double *__restrict__ pre_hi, int G, double *acc)
{
    pmfft::wave_sync();
    for (uint32_t e = lane_id(); e < cnt; e += 64) {
        const uint2 h = list[e];
        const double lo = pre_lo[h.x], hi = pre_hi[h.x];
        lds_add(&acc[h.y], lo);
        lds_add(&acc[(h.y + 1 == (uint32_t)G) ? 0 : h.y + 1], hi);
    }
    pmfft::wave_sync();
}

// Particles [beg, end) of one source row into the LDS row `acc`, 64 per pass, three passes ahead in
// flight (the inner loop is unrolled by four so that a pass stays in the registers it was loaded
// into).  Light particles add their own masses; the first particle of a piece is put on the wave's
// list `heads`, and the listed pieces' sums are fetched and added 64 at a time; the other particles of
// a piece do nothing.  When a pass ends inside a piece that reaches beyond the passes in flight, the
// walk restarts at the end of the piece.
__device__ __forceinline__ void deposit_heavy_run(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key,
                                                  const uint32_t *__restrict__ skip, const double *__restrict__ pre, int64_t cap,
                                                  uint32_t beg, uint32_t end, int di, int dj, int G, double *acc, uint2 *heads)
{
    const int lane = lane_id();
    const double *pre_lo = pre + (int64_t)(di + 2 * dj) * cap, *pre_hi = pre + (int64_t)(di + 2 * dj + 4) * cap;
    uint32_t cnt = 0;
    // returns where to go on: the next pass, or past a long piece
    auto pass = [&](const heavy_t &a, uint32_t base) -> uint32_t {
        const int kc = (int)(a.key % (uint32_t)G);
        int k = 0x7fffffff - lane;
        double lo = 0.0, hi = 0.0;
        const bool light = a.sk == 0u;
        if (light) {
            const double dx = (double)a.f.dx, dy = (double)a.f.dy, dz = (double)a.f.dz;
            const double w = (double)a.f.m * (di ? dx : 1.0 - dx) * (dj ? dy : 1.0 - dy);
            k = kc; lo = w * (1.0 - dz); hi = w * dz;
        }
        add_pass(acc, G, k, lo, hi, light);
        const bool head = (a.sk & SKIP_HEAD) != 0u;
        const unsigned long long hm = __ballot(head);
        if (hm) {
            if (head) heads[cnt + __popcll(hm & ((1ull << lane) - 1ull))] = make_uint2(base + lane, (uint32_t)kc);
            cnt += (uint32_t)__popcll(hm);
            if (cnt >= 64) { flush_heads(heads, cnt, pre_lo, pre_hi, G, acc); cnt = 0; }
        }
        const uint32_t past = __shfl(a.sk, 63) & ~SKIP_HEAD;       // > 1 when the pass ends inside a piece
        return (past >= base + 256 && past > 1u) ? past : base + 64;
    };
    uint32_t base = beg;
    while (base < end) {                                        // wave-uniform throughout
        heavy_t p0 = load_heavy(F, key, skip, base, end), p1 = load_heavy(F, key, skip, base + 64, end),
                p2 = load_heavy(F, key, skip, base + 128, end), p3;
        for (;;) {
            uint32_t next;
            p3 = load_heavy(F, key, skip, base + 192, end);
            next = pass(p0, base);
            if (next != base + 64 || next >= end) { base = next; break; }
            base = next;
            p0 = load_heavy(F, key, skip, base + 192, end);
            next = pass(p1, base);
            if (next != base + 64 || next >= end) { base = next; break; }
            base = next;
            p1 = load_heavy(F, key, skip, base + 192, end);
            next = pass(p2, base);
            if (next != base + 64 || next >= end) { base = next; break; }
            base = next;
            p2 = load_heavy(F, key, skip, base + 192, end);
            next = pass(p3, base);
            if (next != base + 64 || next >= end) { base = next; break; }
            base = next;
        }
    }
    if (cnt) flush_heads(heads, cnt, pre_lo, pre_hi, G, acc);
}

// A summed row leaves the kernel either as G float32 densities (SPECTRUM = false) or, for G = 512,
// already transformed along z: the wave packs the row into 256 complex numbers, runs the 256-point
// FFT and writes the 257 values of the row's half spectrum (SPECTRUM = true).  The density mesh is
// then never written or read; the bits are those of writing it and transforming it afterwards.
template <bool SPECTRUM>
__device__ __forceinline__ void emit_row(double *acc, const float2 *tw, float *__restrict__ density, float2 *__restrict__ spectrum,
                                         uint32_t row, int G, int pitch)
{
    const int j = lane_id();
    if constexpr (SPECTRUM) {
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = j + 64 * r;
            v[r] = make_float2((float)acc[2 * n], (float)acc[2 * n + 1]);
        }
        float2 *buf = (float2 *)acc;                    // the row's LDS is free once every lane holds its part
        pmfft::wave_fft256<-1>(v, buf, tw, j);
        pmfft::wave_r2c_split512(v, buf, tw, j, spectrum + (size_t)row * pitch);
    } else {
        float *out = density + (size_t)row * G;
        for (int c = j; c < G; c += 64) out[c] = (float)acc[c];
    }
}

template <bool SPECTRUM>
__device__ __forceinline__ void emit_empty_row(float *__restrict__ density, float2 *__restrict__ spectrum, uint32_t row, int G, int pitch)
{
    const int j = lane_id();
    if constexpr (SPECTRUM) {
        float2 *dst = spectrum + (size_t)row * pitch;
        for (int c = j; c <= G / 2; c += 64) dst[c] = make_float2(0.0f, 0.0f);
    } else {
        float *out = density + (size_t)row * G;
        for (int c = j; c < G; c += 64) out[c] = 0.0f;
    }
}

// One wave per density row; rows for k_deposit_heavy are only listed.
template <bool SPECTRUM>
__global__ void __launch_bounds__(BLOCK)
k_deposit_rows(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key, const uint8_t *__restrict__ row_flag,
               const uint32_t *__restrict__ rs, float *__restrict__ density,
               float2 *__restrict__ spectrum, const float2 *__restrict__ twiddle,
               uint32_t *__restrict__ heavy_rows, uint32_t *__restrict__ heavy_count, pm_geom g)
{
    extern __shared__ double lds_rows[];                // BLOCK/64 rows of G doubles, then 512 twiddles if SPECTRUM
    const int G = g.G;
    const uint32_t R = (uint32_t)G * (uint32_t)G;
    const uint32_t row = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    float2 *tw = (float2 *)(lds_rows + (size_t)(BLOCK / 64) * G);
    if constexpr (SPECTRUM) {
        for (int q = threadIdx.x; q < 512; q += BLOCK) tw[q] = twiddle[q];
        __syncthreads();                                // the only block-wide barrier: before any wave leaves
    }
    if (row >= R) return;
    double *acc = lds_rows + (size_t)(threadIdx.x >> 6) * G;
    const row_runs t = runs_of(rs, row_flag, row, G);
    if (t.longest > HEAVY_RUN || t.flagged) {
        if (lane_id() == 0) {
            const uint32_t shard = row % ROW_SHARDS;        // holds at most R / ROW_SHARDS + 1 rows
            heavy_rows[(size_t)shard * (R / ROW_SHARDS + 1) + atomicAdd(heavy_count + shard * COUNTER_STRIDE, 1u)] = row;
        }
        return;
    }
    if (t.longest == 0) {
        emit_empty_row<SPECTRUM>(density, spectrum, row, G, g.Ghp);
        return;
    }
    const packed_runs p(t);
    const light_t first = p.load(F, key, 0);
    for (int c = lane_id(); c < G; c += 64) acc[c] = 0.0;
    deposit_light(F, key, p, first, G, acc);
    pmfft::wave_sync();
    emit_row<SPECTRUM>(acc, tw, density, spectrum, row, G, g.Ghp);
}

// Listed rows: a block of 4*S waves; wave w walks slice w/4 of run w%4 into its own LDS row, and the
// rows are added in wave order.
template <bool SPECTRUM>
__global__ void __launch_bounds__(1024)
k_deposit_heavy(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key, const uint32_t *__restrict__ skip,
                const uint8_t *__restrict__ row_flag, const double *__restrict__ pre, int64_t cap, const uint32_t *__restrict__ rs,
                float *__restrict__ density, float2 *__restrict__ spectrum, const float2 *__restrict__ twiddle,
                const uint32_t *__restrict__ heavy_rows, const uint32_t *__restrict__ heavy_count, pm_geom g)
{
    extern __shared__ double lds_rows[];    // per wave a row of G doubles, then per wave a head list, then 512 twiddles if SPECTRUM
    const int G = g.G;
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6;
    const int q = w & 3, slice = w >> 2, slices = nw >> 2;
    double *acc = lds_rows + (size_t)w * G;
    uint2 *heads = (uint2 *)(lds_rows + (size_t)nw * G) + (size_t)w * HEAD_LIST;
    float2 *tw = (float2 *)((uint2 *)(lds_rows + (size_t)nw * G) + (size_t)nw * HEAD_LIST);
    if constexpr (SPECTRUM)
        for (int c = threadIdx.x; c < 512; c += blockDim.x) tw[c] = twiddle[c];
    // blocks s, s + ROW_SHARDS, ... serve shard s
    const uint32_t shard = blockIdx.x % ROW_SHARDS;
    const uint32_t cnt = heavy_count[shard * COUNTER_STRIDE];
    heavy_rows += (size_t)shard * ((uint32_t)G * (uint32_t)G / ROW_SHARDS + 1);
    for (uint32_t e = blockIdx.x / ROW_SHARDS; e < cnt; e += gridDim.x / ROW_SHARDS) {
        const uint32_t row = heavy_rows[e];
        const row_runs t = runs_of(rs, row_flag, row, G);
        for (int c = lane_id(); c < G; c += 64) acc[c] = 0.0;
        // slices are whole multiples of 64 particles
        uint32_t len = t.end[q] - t.beg[q];
        uint32_t sl = ((len + slices - 1) / slices + 63u) & ~63u;
        uint32_t b = t.beg[q] + (uint32_t)slice * sl, e2 = b + sl;
        if (b > t.end[q]) b = t.end[q];
        if (e2 > t.end[q]) e2 = t.end[q];
        deposit_heavy_run(F, key, skip, pre, cap, b, e2, q & 1, q >> 1, G, acc, heads);
        __syncthreads();
        // column c of the block's rows, added in wave order, lands in row 0
        for (int c = threadIdx.x; c < G; c += blockDim.x) {
            double sum = 0.0;
            for (int v = 0; v < nw; ++v) sum += lds_rows[(size_t)v * G + c];
            lds_rows[c] = sum;
        }
        __syncthreads();
        if (w == 0) emit_row<SPECTRUM>(lds_rows, tw, density, spectrum, row, G, g.Ghp);
        __syncthreads();
    }
}

// The rows k_deposit_heavy<.., false> left in a scratch mesh as float32, transformed along z into the
// half spectrum: one wave per listed row.
__global__ void __launch_bounds__(BLOCK)
k_heavy_rows_r2c(const float *__restrict__ rows, float2 *__restrict__ spectrum, const float2 *__restrict__ twiddle,
                 const uint32_t *__restrict__ heavy_rows, const uint32_t *__restrict__ heavy_count, int G, int pitch)
{
    __shared__ float2 tw[512];
    __shared__ float2 bufs[BLOCK / 64][256];
    const int j = lane_id(), w = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < 512; q += BLOCK) tw[q] = twiddle[q];
    __syncthreads();
    const uint32_t shard = blockIdx.x % ROW_SHARDS;
    const uint32_t cnt = heavy_count[shard * COUNTER_STRIDE];
    heavy_rows += (size_t)shard * ((uint32_t)G * (uint32_t)G / ROW_SHARDS + 1);
    const uint32_t waves = (gridDim.x / ROW_SHARDS) * (BLOCK / 64);
    for (uint32_t e = (blockIdx.x / ROW_SHARDS) * (BLOCK / 64) + w; e < cnt; e += waves) {
        const uint32_t row = heavy_rows[e];
        const float2 *src = (const float2 *)(rows + (size_t)row * G);
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = src[j + 64 * r];
        pmfft::wave_fft256<-1>(v, bufs[w], tw, j);
        pmfft::wave_r2c_split512(v, bufs[w], tw, j, spectrum + (size_t)row * pitch);
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

// 4-byte-aligned runs of 2 and 4 floats along a mesh row
struct __attribute__((packed, aligned(4))) f2u { float v[2]; };
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

// corner order of functions.py:61-68 and :140-178: 000 100 010 001 110 101 011 111
__device__ constexpr int DI[8] = {0, 1, 0, 0, 1, 1, 0, 1}, DJ[8] = {0, 0, 1, 0, 1, 0, 1, 1}, DK[8] = {0, 0, 0, 1, 0, 1, 1, 1};

template <typename TP, typename TV, bool FUSED>
__global__ void __launch_bounds__(BLOCK)
k_gather(const pm_prec<TP> *__restrict__ Pin, pm_prec<float> *__restrict__ Pout, const pm_vrec<TV> *__restrict__ Vin,
         pm_vrec<TV> *__restrict__ Vout, const uint32_t *__restrict__ vmap,
         const float *__restrict__ gx, const float *__restrict__ gy, const float *__restrict__ gz,
         const float *__restrict__ phi, float *__restrict__ forces, uint32_t *__restrict__ key_next, int64_t n, pm_geom g,
         double dt, bool integrate)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const int G = g.G;
    const pm_prec<TP> p = Pin[r];
    const cell_t c = locate(p, g);
    const int i1 = (c.i + 1 == G) ? 0 : c.i + 1, j1 = (c.j + 1 == G) ? 0 : c.j + 1, k1 = (c.k + 1 == G) ? 0 : c.k + 1;
    const double wx[2] = {1.0 - c.dx, c.dx}, wy[2] = {1.0 - c.dy, c.dy}, wz[2] = {1.0 - c.dz, c.dz};
    const int ii[2] = {c.i, i1}, jj[2] = {c.j, j1}, kk[2] = {c.k, k1};
    float ga[2][2][2], gb[2][2][2], gc[2][2][2];        // gradient components at corner [di][dj][dk]
    const int64_t s0 = (int64_t)G * G, s1 = G;
    if (FUSED) {
        // np.gradient of the potential, evaluated at the eight corners on the fly
        const bool inner = c.i >= 1 && c.i <= G - 3 && c.j >= 1 && c.j <= G - 3 && c.k >= 1 && c.k <= G - 3;
        if (inner) {
            // all eight corners are interior nodes and nothing wraps: 4 + 4 + 4 short row segments
            const float *b = phi + ((int64_t)c.i * G + c.j) * G + c.k;
            f4u ctr[2][2];                                  // phi(i+a, j+b, k-1 .. k+2)
            f2u xm[2], xp[2], ym[2], yp[2];                 // phi(i-1 | i+2, j+b, k .. k+1), phi(i+a, j-1 | j+2, k .. k+1)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) ctr[a][bb] = *(const f4u *)(b + a * s0 + bb * s1 - 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                xm[q] = *(const f2u *)(b - s0 + q * s1);
                xp[q] = *(const f2u *)(b + 2 * s0 + q * s1);
                ym[q] = *(const f2u *)(b + q * s0 - s1);
                yp[q] = *(const f2u *)(b + q * s0 + 2 * s1);
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    ga[0][bb][cc] = (ctr[1][bb].v[1 + cc] - xm[bb].v[cc]) * 0.5f;
                    ga[1][bb][cc] = (xp[bb].v[cc] - ctr[0][bb].v[1 + cc]) * 0.5f;
                    gb[bb][0][cc] = (ctr[bb][1].v[1 + cc] - ym[bb].v[cc]) * 0.5f;
                    gb[bb][1][cc] = (yp[bb].v[cc] - ctr[bb][0].v[1 + cc]) * 0.5f;
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    gc[a][bb][0] = (ctr[a][bb].v[2] - ctr[a][bb].v[0]) * 0.5f;
                    gc[a][bb][1] = (ctr[a][bb].v[3] - ctr[a][bb].v[1]) * 0.5f;
                }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int a = ii[DI[q]], b = jj[DJ[q]], cc = kk[DK[q]];
                int64_t node = ((int64_t)a * G + b) * G + cc;
                ga[DI[q]][DJ[q]][DK[q]] = diff1(phi, node, s0, a, G);
                gb[DI[q]][DJ[q]][DK[q]] = diff1(phi, node, s1, b, G);
                gc[DI[q]][DJ[q]][DK[q]] = diff1(phi, node, 1, cc, G);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int64_t node = ((int64_t)ii[DI[q]] * G + jj[DJ[q]]) * G + kk[DK[q]];
            ga[DI[q]][DJ[q]][DK[q]] = gx[node]; gb[DI[q]][DJ[q]][DK[q]] = gy[node]; gc[DI[q]][DJ[q]][DK[q]] = gz[node];
        }
    }
    // functions.py:140-178: every add rounds the float32 accumulator
    float fx = 0.0f, fy = 0.0f, fz = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        double w = wx[DI[q]] * wy[DJ[q]] * wz[DK[q]];
        fx = (float)((double)fx + (double)ga[DI[q]][DJ[q]][DK[q]] * w);
        fy = (float)((double)fy + (double)gb[DI[q]][DJ[q]][DK[q]] * w);
        fz = (float)((double)fz + (double)gc[DI[q]][DJ[q]][DK[q]] * w);
    }
    if (forces) { forces[3 * r] = fx; forces[3 * r + 1] = fy; forces[3 * r + 2] = fz; }
    if (!integrate) return;
    // main.py:58-60
    const float dtf = (float)dt;
    float dvx = fx * dtf, dvy = fy * dtf, dvz = fz * dtf;      // float32 product (weak Python float)
    pm_vrec<TV> v = Vin[vmap ? vmap[r] : r];
    // velocities += forces*dt : the add runs in the velocities' dtype
    if constexpr (sizeof(TV) == 4) { v.x = v.x + dvx; v.y = v.y + dvy; v.z = v.z + dvz; }
    else { v.x = v.x + (double)dvx; v.y = v.y + (double)dvy; v.z = v.z + (double)dvz; }
    Vout[r] = v;
    // positions += velocities*dt : float32 product for float32 velocities, the add in the wider
    // of the two dtypes, the result cast to float32 (on store, or by .astype at main.py:60)
    pm_prec<float> o;
    o.m = p.m;
    if constexpr (sizeof(TV) == 4 && sizeof(TP) == 4) {
        o.x = wrap_f32(p.x + v.x * dtf, g.Lf);
        o.y = wrap_f32(p.y + v.y * dtf, g.Lf);
        o.z = wrap_f32(p.z + v.z * dtf, g.Lf);
    } else {
        double mx, my, mz;
        if constexpr (sizeof(TV) == 4) { mx = (double)(v.x * dtf); my = (double)(v.y * dtf); mz = (double)(v.z * dtf); }
        else { mx = v.x * dt; my = v.y * dt; mz = v.z * dt; }
        o.x = wrap_f32((float)((double)p.x + mx), g.Lf);
        o.y = wrap_f32((float)((double)p.y + my), g.Lf);
        o.z = wrap_f32((float)((double)p.z + mz), g.Lf);
    }
    Pout[r] = o;
    // cell key of the new position, for the next step's binning
    const cell_t cn = locate(o, g);
    key_next[r] = (uint32_t)(((int64_t)cn.i * G + cn.j) * G + cn.k);
}

}  // namespace

// ---- launchers -----------------------------------------------------------------------------

hipError_t pm_launch_pack(pm_ctx *c, const void *pos_aos, const float *mass, const void *vel_aos)
{
    const int64_t n = c->n;
    if (n == 0) return hipSuccess;
    const int u = c->cur;
    if (pos_aos) {
        if (c->pos_f64) k_pack_pos<double><<<grid_for(n), BLOCK, 0, c->stream>>>((const double *)pos_aos, mass, c->pd[u], n);
        else k_pack_pos<float><<<grid_for(n), BLOCK, 0, c->stream>>>((const float *)pos_aos, mass, c->pf[u], n);
    } else {
        if (c->vel_f64) k_pack_vel<double><<<grid_for(n), BLOCK, 0, c->stream>>>((const double *)vel_aos, c->vd[c->vcur], n);
        else k_pack_vel<float><<<grid_for(n), BLOCK, 0, c->stream>>>((const float *)vel_aos, c->vf[c->vcur], n);
    }
    return hipGetLastError();
}

hipError_t pm_launch_cell_keys(pm_ctx *c)
{
    int64_t n = c->n;
    if (n == 0) return hipSuccess;
    int u = c->cur;
    if (c->pos_f64) k_cell_keys<double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->pd[u], c->key_in, n, c->g);
    else k_cell_keys<float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->pf[u], c->key_in, n, c->g);
    return hipGetLastError();
}

// Bring the velocity records into the current storage order (they lag one reorder behind).
static hipError_t settle_velocities(pm_ctx *c)
{
    if (!c->v_stale || c->n == 0) { c->v_stale = false; return hipSuccess; }
    int u = c->vcur, w = 1 - u;
    if (c->vel_f64) k_move_v<double><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->vd[u], c->vd[w], c->src, c->n);
    else k_move_v<float><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->vf[u], c->vf[w], c->src, c->n);
    c->vcur = w; c->v_stale = false;
    return hipGetLastError();
}

hipError_t pm_launch_reorder(pm_ctx *c)
{
    int64_t n = c->n;
    hipError_t e = hipMemsetAsync(c->counters, 0, PM_NUM_COUNTERS * sizeof(uint32_t), c->stream);     // heavy rows, heavy-cell pieces
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(c->row_flag, 0, (size_t)c->g.G * c->g.G, c->stream);
    if (e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    int u = c->cur, w = 1 - u;
    const uint32_t cap = pm_piece_shard_cap(c->cap);
    if (c->pos_f64) k_reorder<double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->pd[u], c->pd[w], c->frac, c->src, c->key_out, c->skip, c->row_flag, c->piece_list, c->counters + PM_PIECE_COUNTERS, cap, n, c->g);
    else k_reorder<float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->pf[u], c->pf[w], c->frac, c->src, c->key_out, c->skip, c->row_flag, c->piece_list, c->counters + PM_PIECE_COUNTERS, cap, n, c->g);
    c->cur = w;
    c->v_stale = true;
    return hipGetLastError();
}

hipError_t pm_settle_velocities(pm_ctx *c) { return settle_velocities(c); }

hipError_t pm_launch_row_start(pm_ctx *c)
{
    const uint32_t G = (uint32_t)c->g.G;
    k_row_start<<<grid_for((int64_t)G * G + 1), BLOCK, 0, c->stream>>>(c->key_out, c->n, G, G * G, c->row_start);
    return hipGetLastError();
}

uint32_t pm_piece_shard_cap(int64_t cap)
{
    uint32_t blocks = (uint32_t)((cap + BLOCK - 1) / BLOCK);
    return ((blocks + PIECE_SHARDS - 1) / PIECE_SHARDS) * PIECES_PER_BLOCK;
}

size_t pm_piece_list_entries(int64_t cap) { return (size_t)pm_piece_shard_cap(cap) * PIECE_SHARDS; }
size_t pm_heavy_row_entries(int G) { return ((size_t)G * G / ROW_SHARDS + 1) * ROW_SHARDS; }

static int heavy_waves(int G)
{
    int nw = (int)((128u * 1024u) / ((size_t)G * sizeof(double) + HEAD_LIST * sizeof(uint2))) & ~3;     // a multiple of 4: waves per run
    return nw > PM_HEAVY_WAVES ? PM_HEAVY_WAVES : (nw < 4 ? 4 : nw);
}

template <bool SPECTRUM>
static hipError_t deposit_dispatch(pm_ctx *c)
{
    const int G = c->g.G;
    const uint32_t R = (uint32_t)G * (uint32_t)G;
    hipError_t e = hipMemsetAsync(c->counters, 0, PM_PIECE_COUNTERS * sizeof(uint32_t), c->stream);     // heavy rows; the piece counts stay
    if (e != hipSuccess) return e;
    k_heavy_pieces<<<PIECE_SHARDS * 32, BLOCK, 0, c->stream>>>(c->frac, c->key_out, c->piece_list, c->counters + PM_PIECE_COUNTERS,
                                                             pm_piece_shard_cap(c->cap), c->pre, c->skip, c->n, c->cap);
    const size_t tw_bytes = SPECTRUM ? 512 * sizeof(float2) : 0;
    const size_t lds_light = (size_t)(BLOCK / 64) * G * sizeof(double) + tw_bytes;
    k_deposit_rows<SPECTRUM><<<dim3((R + BLOCK / 64 - 1) / (BLOCK / 64)), BLOCK, lds_light, c->stream>>>(
        c->frac, c->key_out, c->row_flag, c->row_start, c->density, c->spectrum, c->twiddle, c->heavy_rows, c->counters, c->g);
    const int nw = heavy_waves(G);
    const size_t lds_heavy = (size_t)nw * (G * sizeof(double) + HEAD_LIST * sizeof(uint2));
    // heavy rows always leave as float32 rows of the density mesh; in spectrum mode a small kernel transforms them
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute((const void *)k_deposit_heavy<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    k_deposit_heavy<false><<<PM_HEAVY_GRID, nw * 64, lds_heavy, c->stream>>>(c->frac, c->key_out, c->skip, c->row_flag, c->pre, c->cap, c->row_start, c->density,
                                                                   c->spectrum, c->twiddle, c->heavy_rows, c->counters, c->g);
    if (SPECTRUM)
        k_heavy_rows_r2c<<<ROW_SHARDS * 16, BLOCK, 0, c->stream>>>(c->density, c->spectrum, c->twiddle, c->heavy_rows, c->counters, G, c->g.Ghp);
    return hipGetLastError();
}

// to_spectrum: leave the z-transformed rows in the half spectrum instead of the density mesh (G = 512 only)
hipError_t pm_launch_deposit(pm_ctx *c, bool to_spectrum)
{
    return to_spectrum ? deposit_dispatch<true>(c) : deposit_dispatch<false>(c);
}

template <typename TP, typename TV>
static void gather_dispatch(pm_ctx *c, const pm_prec<TP> *P, pm_vrec<TV> *const V[2], double dt, bool integrate, float *forces, bool fused)
{
    int64_t n = c->n;
    pm_prec<float> *out = c->pf[c->cur];
    const pm_vrec<TV> *vin = V[c->vcur];
    pm_vrec<TV> *vout = c->v_stale ? V[1 - c->vcur] : V[c->vcur];
    const uint32_t *vmap = c->v_stale ? c->src : nullptr;
    if (fused)
        k_gather<TP, TV, true><<<grid_for(n), BLOCK, 0, c->stream>>>(P, out, vin, vout, vmap, nullptr, nullptr, nullptr, c->potential, forces, c->key_in, n, c->g, dt, integrate);
    else
        k_gather<TP, TV, false><<<grid_for(n), BLOCK, 0, c->stream>>>(P, out, vin, vout, vmap, c->grad[0], c->grad[1], c->grad[2], nullptr, forces, c->key_in, n, c->g, dt, integrate);
}

hipError_t pm_launch_gather(pm_ctx *c, double dt, bool integrate, bool store_forces, bool fused_gradient)
{
    if (c->n == 0) return hipSuccess;
    int u = c->cur;
    float *f = store_forces ? c->forces : nullptr;
    if (c->pos_f64) {
        if (c->vel_f64) gather_dispatch<double, double>(c, c->pd[u], c->vd, dt, integrate, f, fused_gradient);
        else gather_dispatch<double, float>(c, c->pd[u], c->vf, dt, integrate, f, fused_gradient);
    } else {
        if (c->vel_f64) gather_dispatch<float, double>(c, c->pf[u], c->vd, dt, integrate, f, fused_gradient);
        else gather_dispatch<float, float>(c, c->pf[u], c->vf, dt, integrate, f, fused_gradient);
    }
    if (integrate && c->v_stale) { c->vcur = 1 - c->vcur; c->v_stale = false; }     // the gather moved the velocity records
    return hipGetLastError();
}

hipError_t pm_launch_export_particles(pm_ctx *c, void *pos_aos, void *vel_aos)
{
    int64_t n = c->n;
    if (n == 0) return hipSuccess;
    int u = c->cur, v = c->vcur;
    const uint32_t *vmap = c->v_stale ? c->src : nullptr;
    if (pos_aos) {
        if (c->pos_f64) {
            if (c->vel_f64) k_export_pos<double, double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->pd[u], c->vd[v], vmap, (double *)pos_aos, n);
            else k_export_pos<double, float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->pd[u], c->vf[v], vmap, (double *)pos_aos, n);
        } else {
            if (c->vel_f64) k_export_pos<float, double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->pf[u], c->vd[v], vmap, (float *)pos_aos, n);
            else k_export_pos<float, float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->pf[u], c->vf[v], vmap, (float *)pos_aos, n);
        }
    }
    if (vel_aos) {
        if (c->vel_f64) k_export_vel<double><<<grid_for(n), BLOCK, 0, c->stream>>>(c->vd[v], (double *)vel_aos, n);
        else k_export_vel<float><<<grid_for(n), BLOCK, 0, c->stream>>>(c->vf[v], (float *)vel_aos, n);
    }
    return hipGetLastError();
}

hipError_t pm_launch_export_forces(pm_ctx *c, float *aos)
{
    if (c->n == 0) return hipSuccess;
    const uint32_t *vmap = c->v_stale ? c->src : nullptr;
    if (c->vel_f64) k_export_forces<double><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->forces, c->vd[c->vcur], vmap, aos, c->n);
    else k_export_forces<float><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->forces, c->vf[c->vcur], vmap, aos, c->n);
    return hipGetLastError();
}

hipError_t pm_launch_export_keys(pm_ctx *c, int64_t *keys, int64_t *perm)
{
    if (c->n == 0) return hipSuccess;
    const uint32_t *vmap = c->v_stale ? c->src : nullptr;
    if (c->vel_f64) k_export_keys<double><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->key_out, c->vd[c->vcur], vmap, keys, perm, c->n);
    else k_export_keys<float><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->key_out, c->vf[c->vcur], vmap, keys, perm, c->n);
    return hipGetLastError();
}
