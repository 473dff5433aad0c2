// Particle-side kernels of the PM step for gfx950: cell keys, reorder into cell order, row-start
// table, CIC deposit (row-centric, LDS accumulation, no global atomics), CIC gather fused with
// kick / drift / wrap, and the host-copy helpers.
//
// Arithmetic follows what numba compiles the reference loops to (functions.py:19-70, 99-180):
// wrap, scale, fractional parts and weights in float64; this file is built with
// -ffp-contract=off so no product is fused into an add.
#include "pm_internal.hpp"
#include "pm_fft_dev.hpp"

#include <atomic>
#include <thread>
#include <cstdlib>
#include <type_traits>
#include "pm_cell_dev.hpp"

namespace {

constexpr int BLOCK = 256;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// index of the wave in its block, as a value the compiler knows to be the same in all lanes (so that
// what is computed from it stays in scalar registers and is loaded through the scalar cache)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

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

using pmcell::cell_t;
using pmcell::locate;
using pmcell::cell_key;

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

// The hardware hands workgroup b to XCD b % 8, and every XCD has an L2 of its own.  xcd_block<C>() renumbers the
// workgroups so that each XCD takes C consecutive blocks of every run of 8 C blocks (C = 0: no renumbering).  The
// gather uses C = 16: the eight XCDs still work side by side on one neighbourhood of the mesh (what they read of the
// potential was written last by the previous kernel and sits in the last-level cache), but the mesh rows under 4096
// consecutive particles are fetched into one L2 only.  -3 % for that kernel on the clustered set, -1.5 % on the uniform
// one; one contiguous eighth of the particles per XCD is 30 % slower.
template <int C>
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t blocks)
{
    if (C == 0) return b;
    const uint32_t span = 8u * (uint32_t)(C ? C : 1), base = b / span * span;
    if (base + span > blocks) return b;              // the last, partial run keeps its order
    const uint32_t o = b - base;
    return base + (o & 7u) * (uint32_t)C + (o >> 3);
}
template <int C> __device__ __forceinline__ uint32_t xcd_block() { return xcd_block<C>(blockIdx.x, gridDim.x); }

// Stores and loads of data that is written or read once, front to back, by kernels whose other accesses -- records fetched
// through the permutation, rows of the potential -- are the ones worth keeping in the L2: marked non-temporal (the reorder
// pass's sorted records and fraction records, the gather's records in and out and its keys; the sort's passes read their
// input the same way).  Headline step 1.921 -> 1.897 ms; the same hint on the loads of the column passes or of the inverse
// row pass costs more elsewhere than it gains there (1.95 and 1.90 ms), on the units' particle loads it changes nothing,
// and on the scattered stores of the sort's passes, which the L2 puts together into lines, it costs 125 us.
// Only for sets that do not fit the caches anyway (more than PM_STREAM_MIN particles): on 1M particles, where one kernel's
// output is the next one's cache hit, the same hints cost a step of 0.178 ms 6 us.
#ifndef PM_STREAM_MIN
#define PM_STREAM_MIN (1 << 22)
#endif
typedef uint32_t pm_u4 __attribute__((ext_vector_type(4)));
// (STREAM is a template argument of the kernels, chosen by the launcher: as a run-time condition the two stores are merged
// into one by the compiler and the hint is lost)
template <bool STREAM, typename T> __device__ __forceinline__ void stream_store(T *p, const T &v)
{
    if constexpr (STREAM && sizeof(T) == 16) __builtin_nontemporal_store(*(const pm_u4 *)&v, (pm_u4 *)p);
    else if constexpr (STREAM && sizeof(T) == 4) __builtin_nontemporal_store(*(const uint32_t *)&v, (uint32_t *)p);
    else *p = v;
}
template <bool STREAM, typename T> __device__ __forceinline__ T stream_load(const T *p)
{
    if constexpr (STREAM && sizeof(T) == 16) { const pm_u4 v = __builtin_nontemporal_load((const pm_u4 *)p); return *(const T *)&v; }
    else if constexpr (STREAM && sizeof(T) == 4) { const uint32_t v = __builtin_nontemporal_load((const uint32_t *)p); return *(const T *)&v; }
    else return *p;
}
#ifndef PM_GATHER_XCD
#define PM_GATHER_XCD 16
#endif

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

template <typename TV, typename TF = float>
__global__ void k_export_forces(const TF *__restrict__ in, const pm_vrec<TV> *__restrict__ V, const uint32_t *__restrict__ vmap,
                                TF *__restrict__ out, int64_t n)
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
    key[p] = cell_key(locate(P[p], g), g);
}

// Cell order -> storage order.  Slot r of the new storage takes the position record of old slot
// src[r] and gets a fraction record {dx, dy, dz, mass} -- the particle's offsets inside its cell
// (functions.py:41-43, computed in float64 and rounded to float32 once) -- which is all the deposit
// needs besides the cell key, so that the wrap / scale / floor arithmetic runs once per particle
// here and not once per visit there.  The velocity records are not moved here: the gather, which reads and writes every one of
// them anyway, reads record src[r] and writes it to slot r (k_move_v does it when no gather follows).
//
// Tuning: a density row fed by at most PM_PACKED_MAX particles is walked one particle per lane; a
// mesh row that holds more than PM_HEAVY_SRC particles is deposited on its own, in units of PM_UNIT.
#ifndef PM_PACKED_MAX
#define PM_PACKED_MAX 512
#endif
#ifndef PM_HEAVY_SRC
#define PM_HEAVY_SRC 2048
#endif
#ifndef PM_UNIT
#define PM_UNIT 6144
#endif
#ifndef PM_UNIT_GRID
#define PM_UNIT_GRID 2048
#endif

template <typename TV>
__global__ void k_move_v(const pm_vrec<TV> *__restrict__ Vin, pm_vrec<TV> *__restrict__ Vout, const uint32_t *__restrict__ src, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r < n) Vout[r] = Vin[src[r]];
}

// (the reorder pass renumbers its workgroups like the gather, in runs of 64 per XCD: the records a stretch of the new
// order pulls in come from a few neighbourhoods of the old one, and one L2 then fetches them instead of eight.  Measured
// at 16.7 M particles: 184 us as launched, 171 with runs of 64, 175 with runs of 512, 193 with a contiguous eighth each.)
#ifndef PM_REORDER_XCD
#define PM_REORDER_XCD 64
#endif
// rs_blocks > 0: the first rs_blocks workgroups of the launch are those of k_row_start (a row per thread, one bisection each),
// which like the reorder needs the sorted keys only: their chains of dependent loads run beside the reorder's streams
// instead of after them, in a launch of their own (16.7 M particles: 16 us and a launch less; 1M / 128^3: 5 us).
__device__ __forceinline__ void row_start_of(const uint32_t *__restrict__ key, int64_t n, uint32_t G, uint32_t R, uint32_t *__restrict__ rs, uint32_t row);

template <typename TP, bool STREAM>
__global__ void k_reorder(const pm_prec<TP> *__restrict__ Pin, pm_prec<TP> *__restrict__ Pout,
                          pm_frac *__restrict__ F, const uint32_t *__restrict__ src, int64_t n, pm_geom g,
                          const uint32_t *__restrict__ key, uint32_t R, uint32_t *__restrict__ rs, uint32_t rs_blocks)
{
    if (blockIdx.x < rs_blocks) {
        row_start_of(key, n, (uint32_t)g.G, R, rs, blockIdx.x * BLOCK + threadIdx.x);
        return;
    }
    uint32_t blk = blockIdx.x - rs_blocks;
    const uint32_t blocks = gridDim.x - rs_blocks;
    if constexpr (PM_REORDER_XCD < 0) {                 // every XCD one contiguous eighth of the slots
        const uint32_t per = blocks / 8u;
        if (blk < per * 8u) blk = (blk & 7u) * per + (blk >> 3);
    } else blk = xcd_block<PM_REORDER_XCD>(blk, blocks);
    int64_t r = (int64_t)blk * BLOCK + threadIdx.x;
    if (r >= n) return;
    const pm_prec<TP> rec = Pin[stream_load<STREAM>(src + r)];
    stream_store<STREAM>(Pout + r, rec);
    stream_store<STREAM>(F + r, pmcell::fraction_record(rec, g));
}

// row_start[row] = number of particles whose mesh row (key / G = i*G + j) is < row, for row in
// [0, G*G]: one bisection of the sorted keys per row.
__device__ __forceinline__ void row_start_of(const uint32_t *__restrict__ key, int64_t n, uint32_t G, uint32_t R, uint32_t *__restrict__ rs, uint32_t row)
{
    if (row > R) return;
    const uint64_t target = (uint64_t)row * G;          // first key of the row
    int64_t lo = 0, hi = n;                             // first index in [lo, hi] with key >= target
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)key[mid] < target) lo = mid + 1; else hi = mid;
    }
    rs[row] = (uint32_t)lo;
}
__global__ void k_row_start(const uint32_t *__restrict__ key, int64_t n, uint32_t G, uint32_t R, uint32_t *__restrict__ rs)
{
    row_start_of(key, n, G, R, rs, blockIdx.x * BLOCK + threadIdx.x);
}

// ---- CIC deposit ---------------------------------------------------------------------------
//
// One wave per mesh row (i,j,:) of the density.  The row receives mass from the particles of
// the four mesh rows (i-di, j-dj), di,dj in {0,1}, through the corners (di,dj,0) into cell k and
// (di,dj,1) into cell k+1 of their clouds (functions.py:56-68).  With the particles in cell
// order each of those rows is one contiguous run of the particle array, found through
// row_start.  A wave walks the runs and adds the float64 corner masses (mass * wx * wy * wz left
// to right, weights in float64 from the float32 offsets of the fraction records) into a float64
// copy of the row held in LDS; the row is rounded to float32 once, when it is written out.  No
// global atomics, no mesh-sized index: the only mesh traffic is the one coalesced write of the
// density.
//
// All adds to an LDS row come from one wave, in program order, and the LDS rows are added in a
// fixed order, so the sum has one fixed order.
//
// Two walks.  Rows fed by few particles take their four runs as one packed sequence, one particle
// per lane and pass.  Longer runs are walked 256 particles at a time, each lane taking four
// consecutive particles: in cell order these mostly share a cell, so the lane sums them in
// registers and adds to LDS only when the cell changes; what a lane still holds at the end of the
// four goes through add_pass, which first combines neighbouring lanes that hold the same cell.
//
//
// A crowded mesh row (more than HEAVY_SRC particles) would keep the waves of the four density rows it
// feeds busy long after the others have finished, each of them reading the same particles.  Such a
// source row is cut into units of UNIT particles instead (k_plan_units); k_deposit_units reads every
// unit once and sums it into four float64 rows of its own in global memory, one per (di,dj); the waves
// of the density rows leave crowded sources out, and where a density row has one, its sum is put
// together by k_sum_units: what its wave summed from the other sources, then the units of its crowded
// sources, in a fixed order.
constexpr uint32_t PACKED_MAX = PM_PACKED_MAX, HEAVY_SRC = PM_HEAVY_SRC, UNIT = PM_UNIT, NO_UNIT = 0xffffffffu;
constexpr int UNIT_BLOCK = 128;         // two waves, each with four LDS rows
#ifndef PM_ROW_QUEUES
#define PM_ROW_QUEUES 64
#endif
// counters[]: units handed out, density rows listed (own 128-byte lines), then the turn counters of the row kernel (a line each)
// (COUNTER_PLANNED: workgroups of k_plan_units that are done, where that kernel is a deposit's only one beside the rows')
constexpr int COUNTER_UNITS = 0, COUNTER_PLANNED = 1, COUNTER_ROWS = 32, COUNTER_TICKETS = 64;
constexpr uint32_t COUNTER_WORDS = 64 + PM_ROW_QUEUES * 32;         // words of one set of counters

__device__ __forceinline__ void lds_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifdef PM_TRACE
// Tuning builds only (tools/deposit_trace.py): per wave of the deposit kernels the start and the end of its work on
// the 100 MHz clock, the particles it walked and where it ran.
constexpr uint32_t TRACE_ROWS = 1u << 19;
__device__ uint32_t g_trace[2][TRACE_ROWS][4];
__device__ __forceinline__ void trace_put(int which, uint32_t at, uint64_t t0, uint32_t work)
{
    if (lane_id() == 0 && at < TRACE_ROWS) {
        const uint64_t t1 = wall_clock64();
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_trace[which][at][0] = (uint32_t)t0; g_trace[which][at][1] = (uint32_t)t1; g_trace[which][at][2] = work; g_trace[which][at][3] = hw;
    }
}
#define TRACE_BEGIN const uint64_t trace_t0 = wall_clock64();
#define TRACE_END(which, at, work) trace_put(which, at, trace_t0, work);
#else
#define TRACE_BEGIN
#define TRACE_END(which, at, work)
#endif

// The lanes hold (k, lo, hi): cell and the masses for cells k and k+1 (idle lanes: a k no cell
// has, distinct per lane).  Lanes that share a cell are mostly neighbours (cell order).  Where many
// neighbours share cells, each stretch of equal k is first summed by a segmented scan over the
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

// The same for the four density rows a source row feeds: acc holds them one after the other.
__device__ __forceinline__ void add_pass4(double *acc, int G, int k, double (&lo)[4], double (&hi)[4], bool add)
{
    const int lane = lane_id();
    int kprev = __shfl_up(k, 1);
    const unsigned long long dup = __ballot(lane > 0 && kprev == k);
    if (__popcll(dup) > 8) {
        const unsigned long long starts = ~dup;
        const int start = 63 - __clzll((long long)(starts & ((2ull << lane) - 1ull)));
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const bool take = lane - d >= start;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double lu = __shfl_up(lo[q], d), hu = __shfl_up(hi[q], d);
                if (take) { lo[q] += lu; hi[q] += hu; }
            }
        }
        add = add && (lane == 63 || ((starts >> (lane + 1)) & 1ull));
    }
    if (add) {
        const int k1 = (k + 1 == G) ? 0 : k + 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lds_add(&acc[q * G + k], lo[q]);
            lds_add(&acc[q * G + k1], hi[q]);
        }
    }
}

// beg/end: particle range of the four source rows of a density row, in the order
// (di,dj) = (0,0) (1,0) (0,1) (1,1); key0: the key of cell k = 0 of each source row.
struct row_runs { uint32_t beg[4], end[4], key0[4], unit[4], longest; };    // unit: first unit of a crowded source, or NO_UNIT

__device__ __forceinline__ row_runs runs_of(const uint32_t *__restrict__ rs, const uint32_t *__restrict__ row_unit, uint32_t row, const pm_geom &g)
{
    const int G = g.G;
    int i = (int)(row / (uint32_t)G), j = (int)(row % (uint32_t)G);
    int im = i - 1, jm = (j == 0) ? G - 1 : j - 1;
    // periodic along i, or a slab: plane i - 1 of the first plane and plane i of the extra last plane are not here
    bool has_i = true, has_im = true;
    if (g.Gd == G) { if (im < 0) im = G - 1; }
    else { has_i = i < g.Gs; has_im = i >= 1; }
    uint32_t src[4] = {(uint32_t)(i * G + j), (uint32_t)(im * G + j), (uint32_t)(i * G + jm), (uint32_t)(im * G + jm)};
    row_runs t;
    t.longest = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool has = (q & 1) ? has_im : has_i;
        t.beg[q] = has ? rs[src[q]] : 0u; t.end[q] = has ? rs[src[q] + 1] : 0u;
        t.unit[q] = has ? row_unit[src[q]] : NO_UNIT;
        t.key0[q] = src[q] * (uint32_t)G;
        uint32_t len = t.end[q] - t.beg[q];
        t.longest = len > t.longest ? len : t.longest;
    }
    return t;
}

// The masses a particle adds to cells k and k+1 of the density row that lies (di,dj) from its own.
__device__ __forceinline__ void corner_pair(const pm_frac &f, bool di, bool dj, double &lo, double &hi)
{
    const double dx = (double)f.dx, dy = (double)f.dy, dz = (double)f.dz;
    const double w = (double)f.m * (di ? dx : 1.0 - dx) * (dj ? dy : 1.0 - dy);
    lo = w * (1.0 - dz); hi = w * dz;
}

// ---- packed walk: one particle per lane and pass

struct light_t { uint32_t key; pm_frac f; };

// The four runs as one sequence of `total` particles: entry v is particle slot(v) of run run(v).
struct packed_runs {
    uint32_t o1, o2, o3, total, b0, b1, b2, b3, k0, k1, k2, k3;
    __device__ __forceinline__ explicit packed_runs(const row_runs &t)
    {
        o1 = t.end[0] - t.beg[0]; o2 = o1 + (t.end[1] - t.beg[1]); o3 = o2 + (t.end[2] - t.beg[2]);
        total = o3 + (t.end[3] - t.beg[3]);
        b0 = t.beg[0]; b1 = t.beg[1] - o1; b2 = t.beg[2] - o2; b3 = t.beg[3] - o3;     // modulo 2^32
        k0 = t.key0[0]; k1 = t.key0[1]; k2 = t.key0[2]; k3 = t.key0[3];
    }
    __device__ __forceinline__ int run(uint32_t v) const { return (int)(v >= o1) + (int)(v >= o2) + (int)(v >= o3); }
    __device__ __forceinline__ uint32_t slot(uint32_t v, int q) const { return v + (q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3); }
    __device__ __forceinline__ uint32_t key0(int q) const { return q == 0 ? k0 : q == 1 ? k1 : q == 2 ? k2 : k3; }
    __device__ __forceinline__ light_t load(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key, uint32_t base) const
    {
        light_t t{};
        const uint32_t v = base + lane_id();
        if (v < total) { const uint32_t r = slot(v, run(v)); t.key = key[r]; t.f = F[r]; }
        return t;
    }
};

__device__ __forceinline__ void deposit_packed(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key, const packed_runs &p,
                                               light_t cur, int G, double *acc)
{
    const int lane = lane_id();
    for (uint32_t base = 0; base < p.total; base += 64) {       // wave-uniform
        const uint32_t v = base + lane;
        const pm_frac f = cur.f;
        const uint32_t kk = cur.key;
        if (base + 64 < p.total) cur = p.load(F, key, base + 64);   // in flight while this pass is summed
        int k = 0x7fffffff - lane;
        double lo = 0.0, hi = 0.0;
        const bool live = v < p.total;
        if (live) {
            const int q = p.run(v);
            k = (int)(kk - p.key0(q));
            corner_pair(f, q & 1, q & 2, lo, hi);
        }
        add_pass(acc, G, k, lo, hi, live);
    }
}

// ---- chunked walk: four consecutive particles per lane, 256 per chunk

struct chunk_t { uint4 key; pm_frac f[4]; };

// Chunk at `base` (a multiple of 4): lane l holds particles base + 4 l .. + 3.  A group of four that
// begins before `end` is loaded whole (key[] and frac[] are allocated three entries beyond the
// capacity); what lies outside the run is ignored later.
__device__ __forceinline__ chunk_t load_chunk(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key,
                                              uint32_t base, uint32_t end)
{
    chunk_t c{};
    const uint32_t r = base + 4u * lane_id();
    if (r < end) {
        c.key = *(const uint4 *)(key + r);
#pragma unroll
        for (int t = 0; t < 4; ++t) c.f[t] = F[r + t];
    }
    return c;
}

// Particles [beg, end) of the source row whose cell 0 has key `key0` into the LDS row `acc`.
__device__ __forceinline__ void deposit_chunks(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key,
                                               uint32_t beg, uint32_t end, uint32_t key0, bool di, bool dj,
                                               int G, double *acc)
{
    if (beg >= end) return;
    const int lane = lane_id();
    const int idle = 0x7fffffff - lane;
    uint32_t base = beg & ~3u;
    chunk_t cur = load_chunk(F, key, base, end);
    while (base < end) {                                        // wave-uniform
        const chunk_t c = cur;
        if (base + 256 < end) cur = load_chunk(F, key, base + 256, end);     // in flight while this chunk is summed
        const uint32_t r = base + 4u * lane;
        const uint32_t kk[4] = {c.key.x, c.key.y, c.key.z, c.key.w};
        int ck = idle;
        double slo = 0.0, shi = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool valid = r + t >= beg && r + t < end;
            int k = idle;
            double lo = 0.0, hi = 0.0;
            if (valid) { k = (int)(kk[t] - key0); corner_pair(c.f[t], di, dj, lo, hi); }
            if (k == ck) {
                slo += lo; shi += hi;
            } else {
                if (ck != idle) {                               // the cell has changed: what was summed goes to LDS
                    lds_add(&acc[ck], slo);
                    lds_add(&acc[(ck + 1 == G) ? 0 : ck + 1], shi);
                }
                ck = k; slo = lo; shi = hi;
            }
        }
        add_pass(acc, G, ck, slo, shi, ck != idle);
        base += 256;
    }
}

// Particles [beg, end) of one source row into the four LDS rows acc[q * G ...] of the density rows it feeds,
// q = di + 2 dj: every particle is read once and all eight corner masses are formed from it.
__device__ __forceinline__ void deposit_chunks4(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key,
                                                uint32_t beg, uint32_t end, uint32_t key0, int G, double *acc)
{
    if (beg >= end) return;
    const int lane = lane_id();
    const int idle = 0x7fffffff - lane;
    uint32_t base = beg & ~3u;
    chunk_t cur = load_chunk(F, key, base, end);
    while (base < end) {                                        // wave-uniform
        const chunk_t c = cur;
        if (base + 256 < end) cur = load_chunk(F, key, base + 256, end);
        const uint32_t r = base + 4u * lane;
        const uint32_t kk[4] = {c.key.x, c.key.y, c.key.z, c.key.w};
        int ck = idle;
        double slo[4] = {0.0, 0.0, 0.0, 0.0}, shi[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool valid = r + t >= beg && r + t < end;
            int k = idle;
            double lo[4] = {0.0, 0.0, 0.0, 0.0}, hi[4] = {0.0, 0.0, 0.0, 0.0};
            if (valid) {
                k = (int)(kk[t] - key0);
                // mass * wx * wy * wz, left to right (functions.py:61-68), as corner_pair forms them
                const double dx = (double)c.f[t].dx, dy = (double)c.f[t].dy, dz = (double)c.f[t].dz, m = (double)c.f[t].m;
                const double mx[2] = {m * (1.0 - dx), m * dx}, wy[2] = {1.0 - dy, dy};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double w = mx[q & 1] * wy[q >> 1];
                    lo[q] = w * (1.0 - dz); hi[q] = w * dz;
                }
            }
            if (k == ck) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { slo[q] += lo[q]; shi[q] += hi[q]; }
            } else {
                if (ck != idle) {
                    const int k1 = (ck + 1 == G) ? 0 : ck + 1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { lds_add(&acc[q * G + ck], slo[q]); lds_add(&acc[q * G + k1], shi[q]); }
                }
                ck = k;
#pragma unroll
                for (int q = 0; q < 4; ++q) { slo[q] = lo[q]; shi[q] = hi[q]; }
            }
        }
        add_pass4(acc, G, ck, slo, shi, ck != idle);
        base += 256;
    }
}

// A summed row leaves the kernel either as G float32 densities (NFFT = 0) or, for G = NFFT = 512 or
// 1024, already transformed along z: the wave packs the row into NFFT/2 complex numbers, runs the FFT
// and writes the NFFT/2 + 1 values of the row's half spectrum.  The density mesh is then never written
// or read; the bits are those of writing it and transforming it afterwards.
template <int NFFT>
__device__ __forceinline__ void emit_row(double *acc, const float2 *tw, float *__restrict__ density, float2 *__restrict__ spectrum,
                                         uint32_t row, int G, int pitch, int sb)
{
    const int j = lane_id();
    if constexpr (NFFT != 0) {
        constexpr int E = NFFT / 128;
        float2 v[E];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int n = j + 64 * r;
            v[r] = make_float2((float)acc[2 * n], (float)acc[2 * n + 1]);
        }
        float2 *buf = (float2 *)acc;                    // the row's LDS is free once every lane holds its part
#ifndef PM_DEPOSIT_NOFFT
        pmfft::wave_fft<E, -1>(v, buf, tw, j);
#endif
        pmfft::wave_r2c_split<E>(v, buf, tw, j, spectrum + pm_spec_row(NFFT, sb, (int)(row / NFFT), (int)(row % NFFT)) * pitch);
    } else {
        float *out = density + (size_t)row * G;
        for (int c = j; c < G; c += 64) out[c] = (float)acc[c];
    }
}

template <int NFFT>
__device__ __forceinline__ void emit_empty_row(float *__restrict__ density, float2 *__restrict__ spectrum, uint32_t row, int G, int pitch, int sb,
                                               bool rows_marked)
{
    const int j = lane_id();
    if constexpr (NFFT != 0) {
        if (rows_marked) return;                        // (k_plan_units has marked the row empty: nobody reads it)
        float2 *dst = spectrum + pm_spec_row(NFFT, sb, (int)(row / NFFT), (int)(row % NFFT)) * pitch;
        for (int c = j; c <= G / 2; c += 64) dst[c] = make_float2(0.0f, 0.0f);
    } else {
        float *out = density + (size_t)row * G;
        for (int c = j; c < G; c += 64) out[c] = 0.0f;
    }
}

// Crowded source rows get their units: row_unit[row] = index of the row's first unit (NO_UNIT for the other rows),
// unit_src[u] = the row unit u belongs to.  One counter update per wave.
// row_live (may be null; whole meshes only, G a multiple of 64): a bit per DENSITY row, set where one of the four mesh rows
// that feed it holds a particle.  Where it is written, the deposit writes nothing for the other rows and the transform
// along y that follows takes them as zeros without reading them.
// plan = false: the units are counted but none is planned (row_unit = NO_UNIT everywhere: the row kernel walks every source
// itself, and neither k_deposit_units nor k_sum_units is launched).  The workgroup that finishes last then does the
// housekeeping that is k_sum_units' otherwise: the next deposit's counters start at zero, and the host learns how many units
// there would have been.
__global__ void k_plan_units(const uint32_t *__restrict__ rs, uint32_t R, uint32_t *__restrict__ row_unit, uint32_t *__restrict__ unit_src,
                             uint32_t *__restrict__ counters, unsigned long long *__restrict__ row_live, uint32_t G, bool plan,
                             uint32_t *__restrict__ next_counters, volatile unsigned long long *__restrict__ units_seen, uint32_t turn,
                             unsigned long long *__restrict__ row_need)
{
    const uint32_t row = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t len = row < R ? rs[row + 1] - rs[row] : 0u;
    if (row_need != nullptr) {
        // row_need (with row_live): a bit per POTENTIAL row (a, b), set where the gather of some particle reads the row.  A
        // particle of mesh row (i, j) takes the gradient at the corners (i .. i + 1, j .. j + 1) of its cloud (functions.py:
        // 118-180, wrapped), which np.gradient (main.py:54) differences from the rows one to either side, one-sided at the
        // faces: all of them among (i - 1 .. i + 2, j - 1 .. j + 2), wrapped.  So (a, b) is read if one of the sixteen mesh
        // rows (a - 2 .. a + 1, b - 2 .. b + 1) holds a particle.
        // (G is a multiple of 64 here: a wave holds 64 consecutive b of one a.  Per plane i the wave's lanes look up their
        // own row and three of them the two rows before and the one after the wave's; the marks of b - 2 .. b + 1 are then
        // shifts of one word.)
        const uint32_t lane = (uint32_t)lane_id(), a = row / G, b0 = (row - a * G) & ~63u;
        unsigned long long need = 0ull;
        if (row < R) {
#pragma unroll
            for (uint32_t da = 0; da < 4; ++da) {
                const uint32_t base = ((a + G + 1u - da) % G) * G;
                const uint32_t q = base + b0 + lane;
                const unsigned long long word = __ballot(rs[q + 1] != rs[q]);
                // lanes 0, 1, 2: rows b0 - 2, b0 - 1, b0 + 64 (wrapped)
                const uint32_t e = base + (lane == 2u ? (b0 + 64u) % G : (b0 + G - 2u + lane) % G);
                const unsigned long long edge = __ballot(lane < 3u && rs[e + 1] != rs[e]);
                const unsigned long long m2 = edge & 1ull, m1 = (edge >> 1) & 1ull, p1 = (edge >> 2) & 1ull;
                need |= word | (word << 1) | m1 | (word << 2) | (m1 << 1) | m2 | (word >> 1) | (p1 << 63);
            }
        }
        if (lane == 0 && row < R) row_need[row >> 6] = need;
    }
    if (row_live != nullptr) {
        bool live = len != 0u;
        if (row < R && !live) {
            const uint32_t i = row / G, j = row - i * G;
            const uint32_t im = i == 0u ? G - 1u : i - 1u, jm = j == 0u ? G - 1u : j - 1u;
            const uint32_t a = im * G + j, b = i * G + jm, c = im * G + jm;
            live = rs[a + 1] != rs[a] || rs[b + 1] != rs[b] || rs[c + 1] != rs[c];
        }
        const unsigned long long m = __ballot(live);
        if (lane_id() == 0 && row < R) row_live[row >> 6] = m;
    }
    const uint32_t units = len > HEAVY_SRC ? (len + UNIT - 1) / UNIT : 0u;
    uint32_t before = units;                                    // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(before, d);
        if (lane_id() >= d) before += up;
    }
    const uint32_t total = __shfl(before, 63);
    uint32_t base = 0;
    if (total) {
        if (lane_id() == 0) base = atomicAdd(counters + COUNTER_UNITS, total);
        base = __shfl(base, 0);
    }
    if (plan) {
        if (row >= R) return;
        const uint32_t first = base + before - units;
        row_unit[row] = units ? first : NO_UNIT;
        for (uint32_t u = 0; u < units; ++u) unit_src[first + u] = row;
        return;
    }
    if (row < R) row_unit[row] = NO_UNIT;
    __shared__ uint32_t last;
    __syncthreads();                                    // the workgroup's counts have been added
    if (threadIdx.x == 0) {
        // (two levels -- atomics on one address go one at a time, a thousand of them take as long as the rest of this kernel
        // five times over: the workgroups count themselves in up to 64 groups, on the lines that are the row kernel's turn
        // counters when units are planned, and the last of each group counts the group)
        const uint32_t per = (gridDim.x + 63u) / 64u, group = blockIdx.x / per, groups = (gridDim.x + per - 1u) / per;
        const uint32_t members = (group + 1u) * per <= gridDim.x ? per : gridDim.x - group * per;
        // (No fence: a fence of device scope writes back the L2, which holds what the reorder kernel has just written, and a
        // thousand of those took 30 us.  None is needed: the only thing the last workgroup reads of the others is the unit
        // count, which they add to with atomics of device scope whose results they have used -- so those are done -- before
        // they come here, and which it reads with one.)
        last = 0u;
        if (atomicAdd(counters + COUNTER_TICKETS + 32u * group, 1u) == members - 1u)
            last = atomicAdd(counters + COUNTER_PLANNED, 1u) == groups - 1u;
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0)
        *units_seen = ((unsigned long long)turn << 32) | __hip_atomic_load(counters + COUNTER_UNITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t q = threadIdx.x; q < COUNTER_WORDS; q += BLOCK) next_counters[q] = 0u;
}

// One wave per density row.  A row with a crowded source leaves that source out and is listed for
// k_sum_units: entry = (row, whether the wave has put a sum of the other sources into light_rows[entry]).
// TIGHT: held to 64 registers, eight waves per SIMD instead of seven, at the price of six registers spilled.  On a set
// without crowded rows, where every wave is the same short chain of waits, that is 4 % faster (0.365 -> 0.350 ms on the
// uniform set of 16.7 M particles); on a clustered set 3 % slower.  The launcher chooses.
template <int NFFT, bool TIGHT>
__global__ void __launch_bounds__(BLOCK, TIGHT ? 8 : 1)
k_deposit_rows(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key,
               const uint32_t *__restrict__ rs, const uint32_t *__restrict__ row_unit, float *__restrict__ density,
               float2 *__restrict__ spectrum, const float2 *__restrict__ twiddle,
               uint2 *__restrict__ heavy_rows, uint32_t *__restrict__ counters, double *__restrict__ light_rows, pm_geom g, bool rows_marked)
{
    extern __shared__ double lds_rows[];                // BLOCK/64 rows of G doubles, then the NFFT twiddles
    TRACE_BEGIN
    const int G = g.G;
    const uint32_t R = (uint32_t)g.Gd * (uint32_t)G;
    // last rows first: the reorder pass has just written the records of the last rows, and the transform along y that
    // follows starts with the first planes -- each kernel meets the other's most recent output while caches still hold it
    const uint32_t row = (gridDim.x - 1 - blockIdx.x) * (BLOCK / 64) + (uint32_t)wave_id();
    float2 *tw = (float2 *)(lds_rows + (size_t)(BLOCK / 64) * G);
    double *acc = lds_rows + (size_t)wave_id() * G;
    // A wave's time goes into waiting for what it reads: the twiddle table (the block's copy in LDS), then where its
    // four source runs lie, then the first particles.  All three are requested before the first wait, in this order.
    constexpr int TWQ = NFFT != 0 ? (NFFT + BLOCK - 1) / BLOCK : 1;
    float2 twv[TWQ];
    if constexpr (NFFT != 0) {
#pragma unroll
        for (int q = 0; q < TWQ; ++q) {
            const int at = (int)threadIdx.x + q * BLOCK;
            twv[q] = at < NFFT ? twiddle[at] : make_float2(0.0f, 0.0f);
        }
    }
    const bool in_range = row < R;
    row_runs t{};
    bool crowded = false;
    if (in_range) {
        t = runs_of(rs, row_unit, row, g);
        t.longest = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (t.unit[q] != NO_UNIT) { crowded = true; t.end[q] = t.beg[q]; }     // not walked here
            const uint32_t len = t.end[q] - t.beg[q];
            t.longest = len > t.longest ? len : t.longest;
        }
    }
    const packed_runs p(t);
    const bool walk = in_range && t.longest != 0, packed = p.total <= PACKED_MAX;
    light_t first{};
    if (walk && packed) first = p.load(F, key, 0);
    if (walk)
        for (int c = lane_id(); c < G; c += 64) acc[c] = 0.0;
    if constexpr (NFFT != 0) {
#pragma unroll
        for (int q = 0; q < TWQ; ++q) {
            const int at = (int)threadIdx.x + q * BLOCK;
            if (at < NFFT) tw[at] = twv[q];
        }
        __syncthreads();                                // the only block-wide barrier: before any wave leaves
    }
    if (!in_range) return;
    if (t.longest == 0 && !crowded) {
        emit_empty_row<NFFT>(density, spectrum, row, G, g.Ghp, g.sb, rows_marked);
        TRACE_END(0, row, 0u)
        return;
    }
    uint32_t entry = 0;
    if (crowded) {
        if (lane_id() == 0) {
            entry = atomicAdd(counters + COUNTER_ROWS, 1u);
            heavy_rows[entry] = make_uint2(row, t.longest ? 1u : 0u);
        }
        if (t.longest == 0) return;
        entry = __builtin_amdgcn_readfirstlane(entry);
    }
#ifdef PM_DEPOSIT_NOWALK
    if (false) {} else if (p.total > 100000000u) {
#else
    if (packed) {
        deposit_packed(F, key, p, first, G, acc);
    } else {
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) deposit_chunks(F, key, t.beg[q], t.end[q], t.key0[q], q & 1, q & 2, G, acc);
    }
    pmfft::wave_sync();
    if (crowded) {
        double *out = light_rows + (size_t)entry * G;
        for (int c = lane_id(); c < G; c += 64) out[c] = acc[c];
    } else {
        emit_row<NFFT>(acc, tw, density, spectrum, row, G, g.Ghp, g.sb);
    }
    TRACE_END(0, row, p.total | (crowded ? 0x80000000u : 0u))
}

// The same work by waves that stay: a workgroup loads the twiddles once and each of its waves takes density rows in
// turn until none is left, so that a row costs what is done for it and not the start of a wave, a copy of the twiddle
// table and a barrier as well (with one wave per row those are half of the kernel's time on a clustered set of 16.7 M
// particles: 138 of 281 us with neither walk nor transform).  Rows are handed out one at a time, last rows first, by
// ROW_QUEUES counters: workgroup b uses counter b % ROW_QUEUES and takes the rows r with r % ROW_QUEUES equal to that.
// One row per turn, because consecutive rows are alike and a wave that takes several crowded ones in a row holds the
// kernel up (eight rows per turn: 517 us; four: 394; two: 358; one: 293, against 327 for a wave per row); many
// counters, because atomics on one address go at some 60 million a second, which the 32,768 turns per counter of eight
// counters do not fit into (one row per turn on eight counters: 517 us).  A wave asks for its turn two rows ahead and for
// the places of the next row's runs one row ahead, so that both are there when the row before is done.
constexpr uint32_t ROW_TICKET_STRIDE = 32, ROW_QUEUES = PM_ROW_QUEUES;      // words between two counters (a 128-byte line each)

template <int NFFT>
__global__ void __launch_bounds__(BLOCK)
k_deposit_rows_loop(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key,
                    const uint32_t *__restrict__ rs, const uint32_t *__restrict__ row_unit, float *__restrict__ density,
                    float2 *__restrict__ spectrum, const float2 *__restrict__ twiddle,
                    uint2 *__restrict__ heavy_rows, uint32_t *__restrict__ counters, double *__restrict__ light_rows, pm_geom g, bool rows_marked,
                    uint32_t *__restrict__ tickets)
{
    extern __shared__ double lds_rows[];                // BLOCK/64 rows of G doubles, then the NFFT twiddles
    const int G = g.G;
    const uint32_t R = (uint32_t)g.Gd * (uint32_t)G;
    float2 *tw = (float2 *)(lds_rows + (size_t)(BLOCK / 64) * G);
    double *acc = lds_rows + (size_t)wave_id() * G;
    const uint32_t queues = gridDim.x < ROW_QUEUES ? gridDim.x : ROW_QUEUES, queue = blockIdx.x % queues;
    uint32_t *ticket = tickets + queue * ROW_TICKET_STRIDE;
    // a turn: asked for (lane 0 holds what the counter returns) and, a row later, read
    auto ask = [&]() -> uint32_t { uint32_t t = 0; if (lane_id() == 0) t = atomicAdd(ticket, 1u); return t; };
    // (Measured and not kept: turns 4 t .. 4 t + 3 of a counter as four rows that follow each other, so that the waves that
    // feed on the same mesh rows of particles share an XCD's L2: 355 instead of 293 us on the clustered set.)
    auto turn_of = [&](uint32_t asked) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)asked) * queues + queue; };
    uint32_t asked1 = ask(), asked2 = ask();
    if constexpr (NFFT != 0) {
        for (int q = threadIdx.x; q < NFFT; q += BLOCK) tw[q] = twiddle[q];
        __syncthreads();                                // the only block-wide barrier
    }
    // what is known of a row before its walk
    struct plan_t { row_runs t; bool crowded; };
    auto plan_of = [&](uint32_t row) -> plan_t {
        plan_t q{};
        q.t = runs_of(rs, row_unit, row, g);
        q.t.longest = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (q.t.unit[s] != NO_UNIT) { q.crowded = true; q.t.end[s] = q.t.beg[s]; }     // not walked here
            const uint32_t len = q.t.end[s] - q.t.beg[s];
            q.t.longest = len > q.t.longest ? len : q.t.longest;
        }
        return q;
    };
    uint32_t turn = turn_of(asked1);                    // rows are taken last first: turn u is row R - 1 - u
    plan_t cur{};
    if (turn < R) cur = plan_of(R - 1u - turn);
    while (turn < R) {
        const uint32_t row = R - 1u - turn;
        const uint32_t next_turn = turn_of(asked2);
        asked2 = ask();                                 // (for the row after the next)
        plan_t nxt{};
        if (next_turn < R) nxt = plan_of(R - 1u - next_turn);   // scalar loads on their way during this row's work
        const row_runs &t = cur.t;
        const bool crowded = cur.crowded;
        if (t.longest == 0 && !crowded) {
            emit_empty_row<NFFT>(density, spectrum, row, G, g.Ghp, g.sb, rows_marked);
        } else {
            const packed_runs p(t);
            const bool packed = p.total <= PACKED_MAX;
            uint32_t entry = 0;
            if (crowded && lane_id() == 0) {
                entry = atomicAdd(counters + COUNTER_ROWS, 1u);
                heavy_rows[entry] = make_uint2(row, t.longest ? 1u : 0u);
            }
            if (t.longest != 0) {
                light_t first{};
                if (packed) first = p.load(F, key, 0);
                for (int c = lane_id(); c < G; c += 64) acc[c] = 0.0;
                if (packed) {
                    deposit_packed(F, key, p, first, G, acc);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) deposit_chunks(F, key, t.beg[q], t.end[q], t.key0[q], q & 1, q & 2, G, acc);
                }
                pmfft::wave_sync();
                if (crowded) {
                    entry = __builtin_amdgcn_readfirstlane(entry);
                    double *out = light_rows + (size_t)entry * G;
                    for (int c = lane_id(); c < G; c += 64) out[c] = acc[c];
                } else {
                    emit_row<NFFT>(acc, tw, density, spectrum, row, G, g.Ghp, g.sb);
                }
                pmfft::wave_sync();                     // the row's LDS has been read: the next row may zero it
            }
        }
        cur = nxt;
        turn = next_turn;
    }
}

// One wave per unit of a crowded source row: its particles into four LDS rows, written out as 4 G doubles
// (unit u, corner q = di + 2 dj -> unit_rows[(4 u + q) G ...]).
__global__ void __launch_bounds__(UNIT_BLOCK)
k_deposit_units(const pm_frac *__restrict__ F, const uint32_t *__restrict__ key, const uint32_t *__restrict__ rs,
                const uint32_t *__restrict__ row_unit, const uint32_t *__restrict__ unit_src, const uint32_t *__restrict__ counters,
                double *__restrict__ unit_rows, pm_geom g)
{
    extern __shared__ double lds_rows[];                // per wave four rows of G doubles
    const int G = g.G;
    double *acc = lds_rows + (size_t)wave_id() * 4 * G;
    const uint32_t units = counters[COUNTER_UNITS];
    const uint32_t waves = gridDim.x * (UNIT_BLOCK / 64);
    for (uint32_t u = blockIdx.x * (UNIT_BLOCK / 64) + (uint32_t)wave_id(); u < units; u += waves) {
        TRACE_BEGIN
        const uint32_t row = unit_src[u];
        const uint32_t beg = rs[row] + (u - row_unit[row]) * UNIT;
        const uint32_t last = rs[row + 1], end = beg + UNIT < last ? beg + UNIT : last;
        for (int c = lane_id(); c < 4 * G; c += 64) acc[c] = 0.0;
        deposit_chunks4(F, key, beg, end, row * (uint32_t)G, G, acc);
        pmfft::wave_sync();
        double *out = unit_rows + (size_t)u * 4 * G;
        for (int c = lane_id(); c < 4 * G; c += 64) out[c] = acc[c];
        pmfft::wave_sync();
        TRACE_END(1, u, end - beg)
    }
}

// One block per listed density row: what its own wave summed, then the units of its crowded sources
// (q = 0..3, units in order; a thread per column, four loads in flight), then emitted like any other row.
// threads per block of k_sum_units: a column each at G = 512
#ifndef PM_SUM_BLOCK
#define PM_SUM_BLOCK 512
#endif
constexpr int SUM_BLOCK = PM_SUM_BLOCK;

template <int NFFT>
__global__ void __launch_bounds__(SUM_BLOCK)
k_sum_units(const uint32_t *__restrict__ rs, const uint32_t *__restrict__ row_unit, const uint2 *__restrict__ heavy_rows,
            const uint32_t *__restrict__ counters, const double *__restrict__ unit_rows, const double *__restrict__ light_rows,
            float *__restrict__ density, float2 *__restrict__ spectrum, const float2 *__restrict__ twiddle, pm_geom g,
            uint32_t *__restrict__ next_counters, volatile unsigned long long *__restrict__ units_seen, uint32_t turn)
{
    extern __shared__ double lds_rows[];                // one row of G doubles, then the NFFT twiddles
    const int G = g.G;
    float2 *tw = (float2 *)(lds_rows + G);
    if constexpr (NFFT != 0) {
        for (int q = threadIdx.x; q < NFFT; q += SUM_BLOCK) tw[q] = twiddle[q];
    }
    // housekeeping of the last kernel of a deposit: the next deposit's counters (the other set: nobody uses it now)
    // start at zero, and the host learns how many units this one had (a word of page-locked host memory)
    if (blockIdx.x == 0) {
        for (uint32_t q = threadIdx.x; q < COUNTER_WORDS; q += SUM_BLOCK) next_counters[q] = 0u;
        if (threadIdx.x == 0) *units_seen = ((unsigned long long)turn << 32) | counters[COUNTER_UNITS];
    }
    const uint32_t entries = counters[COUNTER_ROWS];
    for (uint32_t e = blockIdx.x; e < entries; e += gridDim.x) {
        const uint2 entry = heavy_rows[e];
        const row_runs t = runs_of(rs, row_unit, entry.x, g);
        for (int c = threadIdx.x; c < G; c += SUM_BLOCK) {
            double sum = entry.y ? light_rows[(size_t)e * G + c] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (t.unit[q] == NO_UNIT) continue;
                const uint32_t units = (t.end[q] - t.beg[q] + UNIT - 1) / UNIT;
                const size_t step = (size_t)4 * G;
                const double *in = unit_rows + ((size_t)t.unit[q] * 4 + q) * G + c;
                uint32_t u = 0;
#ifndef PM_SUM_NO16
                for (; u + 16 <= units; u += 16, in += 16 * step) {
                    double v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = in[k * step];
#pragma unroll
                    for (int k = 0; k < 16; ++k) sum += v[k];
                }
#endif
#ifndef PM_SUM_NO8
                // sixteen / eight loads in flight for the long runs of units (the adds keep their order)
                for (; u + 8 <= units; u += 8, in += 8 * step) {
                    const double v0 = in[0], v1 = in[step], v2 = in[2 * step], v3 = in[3 * step];
                    const double v4 = in[4 * step], v5 = in[5 * step], v6 = in[6 * step], v7 = in[7 * step];
                    sum += v0; sum += v1; sum += v2; sum += v3; sum += v4; sum += v5; sum += v6; sum += v7;
                }
#endif
                for (; u + 4 <= units; u += 4, in += 4 * step) {
                    const double v0 = in[0], v1 = in[step], v2 = in[2 * step], v3 = in[3 * step];
                    sum += v0; sum += v1; sum += v2; sum += v3;
                }
                for (; u < units; ++u, in += step) sum += in[0];
            }
            lds_rows[c] = sum;
        }
        __syncthreads();
        if (threadIdx.x < 64) emit_row<NFFT>(lds_rows, tw, density, spectrum, entry.x, G, g.Ghp, g.sb);
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

// 4-byte-aligned runs of 2 and 4 floats along a mesh row
struct __attribute__((packed, aligned(4))) f2u { float v[2]; };
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

// corner order of functions.py:61-68 and :140-178: 000 100 010 001 110 101 011 111
__device__ constexpr int DI[8] = {0, 1, 0, 0, 1, 1, 0, 1}, DJ[8] = {0, 0, 1, 0, 1, 0, 1, 1}, DK[8] = {0, 0, 0, 1, 0, 1, 1, 1};

// Where the particles that leave a slab go (SLAB instantiation of k_gather only).
// count[0]: entries of the outbox, count[1]: entries that did not fit, count[2 + d]: entries bound for slab d,
// count[2 + nslabs]: set (and never cleared) once a particle was found in a slab that does not own its cell
// cuts[0 .. nranks]: first plane of every rank's window of planes (cuts[nranks] = G): a leaver goes to the rank whose window
// holds its new cell
struct outbox_t { pm_migrant *rec; uint64_t *key; uint32_t *count; uint32_t cap; const int *cuts; int nranks; };

// the rank r with cuts[r] <= plane < cuts[r + 1]
__device__ __forceinline__ uint32_t rank_of_plane(const outbox_t &ob, int plane)
{
    int lo = 0, hi = ob.nranks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ob.cuts[mid] <= plane) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}

// A wave's leavers (`leaves`: this lane's particle goes; cn_i: plane of its new cell) into the outbox: one counter update
// per wave for the slots, one per wave and destination for the counts.
__device__ __forceinline__ void put_in_outbox(const outbox_t &ob, bool leaves, int cn_i, const pm_migrant &m)
{
    const unsigned long long lm = __ballot(leaves);
    if (!lm) return;
    const int leader = __ffsll((long long)lm) - 1;
    uint32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(ob.count, (uint32_t)__popcll(lm));
    base = __shfl(base, leader);
    uint32_t dest = 0;
    bool kept = false;
    if (leaves) {
        dest = rank_of_plane(ob, cn_i);
        const uint32_t slot = base + (uint32_t)__popcll(lm & ((1ull << lane_id()) - 1ull));
        kept = slot < ob.cap;
        if (kept) {
            ob.rec[slot] = m;
            ob.key[slot] = ((uint64_t)dest << 32) | m.id;
        } else {
            atomicAdd(ob.count + 1, 1u);
        }
    }
    // how many go where: one counter update per wave and destination (a wave's leavers mostly share one)
    unsigned long long todo = __ballot(kept);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const uint32_t d = (uint32_t)__shfl((int)dest, first);
        const unsigned long long same = __ballot(kept && dest == d);
        if (lane_id() == first) atomicAdd(ob.count + 2 + d, (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// SLAB: the context owns planes [i0, i0 + Gs) of the mesh; phi holds them with one plane below and two
// above (plane a of the mesh is plane a - i0 + 1 of the buffer, modulo G), and a particle whose new
// cell lies in another slab is copied to the outbox and gets the key that sorts it behind the rest.
// MAPPED: the velocity record of slot r is record vmap[r] of Vin (the records lag one reorder behind)
// (The float32 instantiations are held to 64 registers, eight waves per SIMD instead of six: the kernel waits for the
// potential, and more waves wait side by side -- 4 % on a uniform particle set, nothing spilled.)
template <typename TP, typename TV, bool FUSED, bool SLAB, bool MAPPED, bool STREAM = false>
__global__ void __launch_bounds__(BLOCK, (sizeof(TP) == 4 && sizeof(TV) == 4) ? 8 : 1)
k_gather(const pm_prec<TP> *__restrict__ Pin, pm_prec<float> *__restrict__ Pout, const pm_vrec<TV> *__restrict__ Vin,
         pm_vrec<TV> *__restrict__ Vout, const uint32_t *__restrict__ vmap,
         const float *__restrict__ gx, const float *__restrict__ gy, const float *__restrict__ gz,
         const float *__restrict__ phi, float *__restrict__ forces, uint32_t *__restrict__ key_next, int64_t n, pm_geom g,
         double dt, bool integrate, outbox_t ob, pm_mover_probe probe)
{
    // (Measured and not kept: wave w of the workgroup taking the w-th sixteen of each of the workgroup's four runs of 64
    // slots -- on a uniform set a quarter of the extent of four neighbouring mesh rows, which share their potential rows,
    // instead of one row from end to end: fewer lines per wave, but 0.62 ms against 0.59, and 3 % lost on a clustered set.)
#ifdef PM_GATHER_BLOCKED
    // (Tuning build: what a cell key blocked in (i, j) would do for this kernel on a UNIFORM set, without changing the key.
    // There a workgroup's 256 slots are about four consecutive mesh rows and a plane is n / G / 256 workgroups, so taking the
    // workgroups of B x B neighbouring (plane, position in plane) places one after the other puts the same neighbouring rows
    // under neighbouring workgroups as that key would.)
    uint32_t wg = blockIdx.x;
    {
        constexpr uint32_t B = PM_GATHER_BLOCKED;
        const uint32_t per_plane = (uint32_t)(n / g.G / BLOCK), planes = (uint32_t)g.G;
        if (per_plane % B == 0 && planes % B == 0 && wg < per_plane * planes) {
            const uint32_t t = wg / (B * B), o = wg % (B * B), gi = t / (per_plane / B), gj = t % (per_plane / B);
            wg = (gi * B + o / B) * per_plane + gj * B + o % B;
        }
    }
    int64_t r = (int64_t)wg * BLOCK + threadIdx.x;
#else
    int64_t r = (int64_t)xcd_block<PM_GATHER_XCD>() * BLOCK + threadIdx.x;
#endif
    if (r >= n) return;
    const int G = g.G;
    const pm_prec<TP> p = stream_load<STREAM>(Pin + r);
    // the velocity record is asked for now -- its slot first, while the position is on its way -- and is there by the
    // time the forces are summed: two trips to memory in a row less than asking for it where it is used
    pm_vrec<TV> v = Vin[MAPPED ? stream_load<STREAM>(vmap + r) : r];
    cell_t c = locate(p, g);
    if constexpr (SLAB) {
        // A particle whose cell lies in another slab should not be here (pm_slab_inbox was handed a record that belongs
        // elsewhere, or the outbox was never emptied).  It is noted in a word that is never cleared -- the caller sees
        // it with the next counts -- and read against the slab's first plane, so that no access leaves the buffers.
        const int il = c.i - g.i0;
        if (g.Gs != G && (il < 0 || il >= g.Gs)) {
            atomicOr(ob.count + 2 + ob.nranks, 1u);
            c.i = g.i0;
        }
    }
    const uint32_t k_old = cell_key(c, g);              // (kept for the count of movers at the end: one register instead of the cell's three)
    const int i1 = (c.i + 1 == G) ? 0 : c.i + 1, j1 = (c.j + 1 == G) ? 0 : c.j + 1, k1 = (c.k + 1 == G) ? 0 : c.k + 1;
    const double wx[2] = {1.0 - c.dx, c.dx}, wy[2] = {1.0 - c.dy, c.dy}, wz[2] = {1.0 - c.dz, c.dz};
    const int ii[2] = {c.i, i1}, jj[2] = {c.j, j1}, kk[2] = {c.k, k1};
    float ga[2][2][2], gb[2][2][2], gc[2][2][2];        // gradient components at corner [di][dj][dk]
    const int64_t s0 = (int64_t)G * G, s1 = G;
    // plane of the potential buffer that holds mesh plane a
    auto plane = [&](int a) {
        if constexpr (!SLAB) return a;
        int l = a - g.i0 + g.lo;
        if (l < 0) l += G; else if (l > g.Gs + 2) l -= G;       // the buffer holds planes 0 .. Gs + 2
        return l;
    };
    if (FUSED) {
        // np.gradient of the potential, evaluated at the eight corners on the fly
        const bool inner = c.i >= 1 && c.i <= G - 3 && c.j >= 1 && c.j <= G - 3 && c.k >= 1 && c.k <= G - 3;
        if (inner) {
            // all eight corners are interior nodes and nothing wraps: 4 + 4 + 4 short row segments
            const float *b = phi + ((int64_t)plane(c.i) * G + c.j) * G + c.k;
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
                int64_t node = ((int64_t)plane(a) * G + b) * G + cc;
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
    // velocities += forces*dt : the add runs in the velocities' dtype
    if constexpr (sizeof(TV) == 4) { v.x = v.x + dvx; v.y = v.y + dvy; v.z = v.z + dvz; }
    else { v.x = v.x + (double)dvx; v.y = v.y + (double)dvy; v.z = v.z + (double)dvz; }
    stream_store<STREAM>(Vout + r, v);
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
    stream_store<STREAM>(Pout + r, o);
    // cell key of the new position, for the next step's binning
    const cell_t cn = locate(o, g);
    const uint32_t kn = cell_key(cn, g);
    stream_store<STREAM>(key_next + r, kn);
    {
        // Which particles changed their cell: a bit per slot for the next binning, which may start from this order.
        // And how many, from a sample of the workgroups (full ones only), for the host to choose that binning by: one counter
        // update per wave; the wave that reports last hands the estimate over.
        const unsigned long long moved = __ballot(kn != k_old);
        if (probe.mask != nullptr && lane_id() == 0) probe.mask[r >> 6] = moved;
        const uint32_t blk = (uint32_t)(r / BLOCK);
        if (probe.expect != 0u && (blk & (probe.every - 1u)) == 0u && (int64_t)(blk + 1u) * BLOCK <= n) {
            if (lane_id() == 0) {
                // one word holds both the waves that have reported (upper half) and their movers (lower half): the
                // wave that completes the count has the sum in hand, and no fence is needed to order two words
                const unsigned long long mine = (1ull << 32) | (unsigned long long)__popcll(moved);
                const unsigned long long all = atomicAdd(probe.count, mine) + mine;
                if ((uint32_t)(all >> 32) == probe.expect) {
                    atomicExch(probe.count, 0ull);
                    const uint32_t seen = (uint32_t)all;
                    *probe.seen = (unsigned long long)seen * probe.every > 0xfffffffeull ? 0xfffffffeu : seen * probe.every;
                }
            }
        }
    }
    if constexpr (SLAB && sizeof(TV) == 4) {
        const bool leaves = kn == (uint32_t)g.Gs * (uint32_t)G * (uint32_t)G;
        put_in_outbox(ob, leaves, cn.i, pm_migrant{o.x, o.y, o.z, o.m, v.x, v.y, v.z, v.id});
    }
}

// ---- slab decomposition: particles entering and leaving -----------------------------------------

__global__ void k_set_ids(pm_vrec<float> *__restrict__ V, const uint32_t *__restrict__ ids, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r < n) V[r].id = ids[r];
}

// outbox entries in sorted order -> one contiguous run of records
__global__ void k_pack_outbox(const pm_migrant *__restrict__ box, const uint32_t *__restrict__ idx, pm_migrant *__restrict__ dst, uint32_t count)
{
    uint32_t r = blockIdx.x * BLOCK + threadIdx.x;
    if (r < count) dst[r] = box[idx[r]];
}

// offsets[d] = first sorted outbox entry bound for slab d or a later one, d = 0 .. nslabs
__global__ void k_outbox_offsets(const uint64_t *__restrict__ key, uint32_t count, int nslabs, uint32_t *__restrict__ offsets)
{
    int d = blockIdx.x * BLOCK + threadIdx.x;
    if (d > nslabs) return;
    const uint64_t target = (uint64_t)d << 32;
    uint32_t lo = 0, hi = count;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (key[mid] < target) lo = mid + 1; else hi = mid;
    }
    offsets[d] = lo;
}

// After the window of a slab has changed: the key of every live particle under the new window; the particles outside
// it go to the outbox, and get the key that sorts them behind the rest, like the leavers of a drift.
__global__ void k_rekey(const pm_prec<float> *__restrict__ P, const pm_vrec<float> *__restrict__ V, uint32_t *__restrict__ key, int64_t n,
                        pm_geom g, outbox_t ob)
{
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = r < n;
    pm_prec<float> p{};
    pm_vrec<float> v{};
    cell_t c{};
    bool leaves = false;
    if (live) {
        p = P[r]; v = V[r];
        c = locate(p, g);
        const uint32_t k = cell_key(c, g);
        key[r] = k;
        leaves = g.Gs != g.G && k == (uint32_t)g.Gs * (uint32_t)g.G * (uint32_t)g.G;
    }
    put_in_outbox(ob, leaves, c.i, pm_migrant{p.x, p.y, p.z, p.m, v.x, v.y, v.z, v.id});
}

// particles per mesh plane, added to counts[G]
__global__ void k_plane_counts(const pm_prec<float> *__restrict__ P, int64_t n, pm_geom g, unsigned long long *__restrict__ counts)
{
    extern __shared__ uint32_t planes[];                // G counters
    for (int q = threadIdx.x; q < g.G; q += BLOCK) planes[q] = 0u;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * BLOCK)
        atomicAdd(planes + locate(P[r], g).i, 1u);
    __syncthreads();
    for (int q = threadIdx.x; q < g.G; q += BLOCK)
        if (planes[q]) atomicAdd(counts + q, (unsigned long long)planes[q]);
}

// arrivals behind the live set, with their cell keys
__global__ void k_append(const pm_migrant *__restrict__ src, pm_prec<float> *__restrict__ P, pm_vrec<float> *__restrict__ V,
                         uint32_t *__restrict__ key, int64_t at, int64_t count, pm_geom g)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= count) return;
    const pm_migrant m = src[r];
    const pm_prec<float> p{m.x, m.y, m.z, m.m};
    P[at + r] = p;
    V[at + r] = pm_vrec<float>{m.vx, m.vy, m.vz, m.id};
    key[at + r] = cell_key(locate(p, g), g);
}

// per-destination counts as 64-bit integers, counts[d], d < nslabs; then counts[nslabs] = the overflow count, plus
// 2^32 if a particle was ever found in a slab that does not own its cell
__global__ void k_outbox_counts(const uint32_t *__restrict__ count, int nslabs, int64_t *__restrict__ out)
{
    const int d = blockIdx.x * BLOCK + threadIdx.x;
    if (d < nslabs) out[d] = (int64_t)count[2 + d];
    else if (d == nslabs) out[d] = (int64_t)count[1] + ((int64_t)count[2 + nslabs] << 32);
}

__global__ void k_export_slab(const pm_prec<float> *__restrict__ P, const pm_vrec<float> *__restrict__ V, const uint32_t *__restrict__ vmap,
                              uint32_t *__restrict__ ids, float *__restrict__ pos, float *__restrict__ vel, int64_t n)
{
    int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const pm_prec<float> p = P[r];
    const pm_vrec<float> v = V[vmap ? vmap[r] : r];
    ids[r] = v.id;
    pos[3 * r] = p.x; pos[3 * r + 1] = p.y; pos[3 * r + 2] = p.z;
    vel[3 * r] = v.x; vel[3 * r + 1] = v.y; vel[3 * r + 2] = v.z;
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

// row_starts: row_start is brought up to date as well (k_row_start's workgroups ride along at the front of the launch)
hipError_t pm_launch_reorder(pm_ctx *c, bool row_starts)
{
    int64_t n = c->n;
    if (n == 0) return row_starts ? pm_launch_row_start(c) : hipSuccess;
    int u = c->cur, w = 1 - u;
    pm_timer timed(c, PM_NSTAGES + PM_KERNEL_REORDER);
    const uint32_t R = (uint32_t)c->g.Gs * (uint32_t)c->g.G;        // mesh rows of the owned planes, as pm_launch_row_start
    const uint32_t rs_blocks = row_starts ? (uint32_t)grid_for((int64_t)R + 1).x : 0u;
    dim3 grid = grid_for(n);
    grid.x += rs_blocks;
    if (c->pos_f64) k_reorder<double, false><<<grid, BLOCK, 0, c->stream>>>(c->pd[u], c->pd[w], c->frac, c->src, n, c->g, c->key_out, R, c->row_start, rs_blocks);
    else if (n > PM_STREAM_MIN) k_reorder<float, true><<<grid, BLOCK, 0, c->stream>>>(c->pf[u], c->pf[w], c->frac, c->src, n, c->g, c->key_out, R, c->row_start, rs_blocks);
    else k_reorder<float, false><<<grid, BLOCK, 0, c->stream>>>(c->pf[u], c->pf[w], c->frac, c->src, n, c->g, c->key_out, R, c->row_start, rs_blocks);
    c->cur = w;
    c->v_stale = true;
    return hipGetLastError();
}

hipError_t pm_settle_velocities(pm_ctx *c) { return settle_velocities(c); }

hipError_t pm_launch_row_start(pm_ctx *c)
{
    const uint32_t G = (uint32_t)c->g.G, R = (uint32_t)c->g.Gs * G;      // mesh rows of the owned planes
    k_row_start<<<grid_for((int64_t)R + 1), BLOCK, 0, c->stream>>>(c->key_out, c->n, G, R, c->row_start);
    return hipGetLastError();
}

// A crowded source row holds more than HEAVY_SRC particles; it has one unit per UNIT particles and a partly
// filled one; it feeds four density rows.
size_t pm_max_units(int64_t cap) { return (size_t)(cap / UNIT + cap / HEAVY_SRC + 1); }
size_t pm_heavy_row_entries(int G, int64_t cap)
{
    const size_t by_particles = (size_t)(4 * (cap / HEAVY_SRC) + 4), rows = (size_t)(G + 1) * G;
    return by_particles < rows ? by_particles : rows;
}

#ifndef PM_DEPOSIT_LOOP_PER_CU
#define PM_DEPOSIT_LOOP_PER_CU 8
#endif
template <int NFFT>
static hipError_t deposit_dispatch(pm_ctx *c)
{
    const int G = c->g.G;
    const uint32_t R = (uint32_t)c->g.Gd * (uint32_t)G, Rs = (uint32_t)c->g.Gs * (uint32_t)G;
    // two sets of counters take turns: each deposit's last kernel zeroes the set of the next one
    uint32_t *counters = c->counters + COUNTER_WORDS * c->counter_set, *next_counters = c->counters + COUNTER_WORDS * (1 - c->counter_set);
    c->counter_set = 1 - c->counter_set;
    hipError_t e = hipSuccess;
    // (a whole mesh whose rows leave as half-spectrum rows for the pass along y of this library: the empty rows are marked
    // instead of written)
    // (grid sizes whose pass along y has a thread read rows jl, jl + 64, ...: one word of the marks per row read)
    const bool mark_rows = (NFFT == 512 || NFFT == 1024) && !c->slab && c->row_live != nullptr && c->row_need != nullptr;
    c->rows_marked = mark_rows;
    // Crowded rows are planned as units only while the deposit before the last had some (pm_ctx::units_seen): a set without
    // them -- a uniform one -- then runs two kernels instead of four.  Should crowded rows appear, the row kernel walks them itself, slowly but with the same sums, and the count
    // that comes back switches the units on for the deposits that follow (a new particle set starts with them on).
    // (PM_DEPOSIT_UNITS in the environment, read at every deposit: "0" never plans units, anything else always -- tests)
    const char *forced = std::getenv("PM_DEPOSIT_UNITS");
    const uint64_t turn = c->deposits++;
    volatile unsigned long long *seen_dev = c->units_seen_dev + (turn & 3u);
    bool planned = true;                                // (the first two deposits of a particle set)
    if (c->deposits_of_set++ >= 2) {
        // the word of the deposit before the last: there since a step ago (a host that is further ahead of the device than
        // that waits here, first politely, then for the stream)
        volatile unsigned long long *word = c->units_seen + ((turn - 2) & 3u);
        const unsigned long long want = (turn - 2 + 1) & 0xffffffffull;
        unsigned long long v = *word;
        for (int spin = 0; (v >> 32) != want && spin < (1 << 16); ++spin) { std::this_thread::yield(); v = *word; }
        if ((v >> 32) != want) {
            if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
            v = *word;
        }
        planned = (v >> 32) != want || (uint32_t)v != 0u;
    }
    if (forced) planned = forced[0] != '0';
    k_plan_units<<<grid_for(Rs), BLOCK, 0, c->stream>>>(c->row_start, Rs, c->row_unit, c->unit_src, counters, mark_rows ? c->row_live : nullptr, (uint32_t)G,
                                                       planned, next_counters, seen_dev, (uint32_t)(turn + 1), mark_rows ? c->row_need : nullptr);
    const size_t tw_bytes = (size_t)NFFT * sizeof(float2);
    const size_t lds_rows = (size_t)(BLOCK / 64) * G * sizeof(double);
    const size_t lds_units = (size_t)(UNIT_BLOCK / 64) * 4 * G * sizeof(double);
    if (lds_units > 64 * 1024) {
        // more LDS than a kernel gets by default: allowed once per device (a bit per device; two threads that both
        // find it unset both set it, which is harmless)
        static std::atomic<uint64_t> allowed{0};
        const uint64_t bit = 1ull << (c->device & 63);
        if (!(allowed.load(std::memory_order_acquire) & bit)) {
            e = hipFuncSetAttribute((const void *)k_deposit_units, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_units);
            if (e != hipSuccess) return e;
            allowed.fetch_or(bit, std::memory_order_release);
        }
    }
    if (lds_rows + tw_bytes > 64 * 1024) {
        static std::atomic<uint64_t> allowed_rows{0};
        const uint64_t bit = 1ull << (c->device & 63);
        if (!(allowed_rows.load(std::memory_order_acquire) & bit)) {
            e = hipFuncSetAttribute((const void *)k_deposit_rows<NFFT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_rows + tw_bytes));
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void *)k_deposit_rows<NFFT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_rows + tw_bytes));
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void *)k_deposit_rows_loop<NFFT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_rows + tw_bytes));
            if (e != hipSuccess) return e;
            allowed_rows.fetch_or(bit, std::memory_order_release);
        }
    }
    // The units need the plan only: they are summed on the side stream while the rows are walked on this one.
#ifdef PM_NO_BESIDE
    const bool beside = false;                          // (tuning builds: each kernel's own duration)
#else
    const bool beside = planned;
#endif
    hipStream_t us = beside ? c->side : c->stream;
    if (beside) {
        if ((e = hipEventRecord(c->ev_fork, c->stream)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(c->side, c->ev_fork, 0)) != hipSuccess) return e;
    }
    if (planned)
        k_deposit_units<<<PM_UNIT_GRID, UNIT_BLOCK, lds_units, us>>>(c->frac, c->key_out, c->row_start, c->row_unit, c->unit_src, counters,
                                                                      c->unit_rows, c->g);
    if (beside && (e = hipEventRecord(c->ev_join, c->side)) != hipSuccess) return e;
    // Two row kernels.  A set with crowded rows (the units run beside) takes the one whose waves stay and take rows in turn:
    // its rows differ a hundredfold in what they cost, and half of a wave-per-row kernel's time is the start of waves that
    // then find little to do (headline set: step 2.00 -> 1.97 ms).  A set without crowded rows takes a wave per row, held to
    // 64 registers: all its rows are alike, eight waves per SIMD hide their waits, and the looping kernel at that register
    // budget spills (uniform set of the same size: 354 us against 506).  The word that decides is a step or two old, as
    // for the units: either kernel is right for any set.
#ifdef PM_DEPOSIT_ONESHOT
    const bool looping = false;
#else
    const bool looping = planned;
#endif
    {
        pm_timer timed(c, PM_NSTAGES + PM_KERNEL_DEPOSIT_ROWS);
        if (looping) {
            // as many workgroups as the GPU holds at once (the LDS allows eight per CU), or fewer for a small mesh
            static int cus[64] = {};
            int &ncu = cus[c->device & 63];
            if (ncu == 0 && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) ncu = 256;
            const uint32_t want = (R + (BLOCK / 64) - 1) / (BLOCK / 64), most = (uint32_t)ncu * PM_DEPOSIT_LOOP_PER_CU;
            const dim3 grid(want < most ? (want ? want : 1u) : most);
            k_deposit_rows_loop<NFFT><<<grid, BLOCK, lds_rows + tw_bytes, c->stream>>>(
                c->frac, c->key_out, c->row_start, c->row_unit, c->density, c->spectrum, c->twiddle, c->heavy_rows, counters, c->light_rows, c->g, mark_rows,
                counters + COUNTER_TICKETS);
        } else {
            const dim3 grid((R + BLOCK / 64 - 1) / (BLOCK / 64));
            if (planned)
                k_deposit_rows<NFFT, false><<<grid, BLOCK, lds_rows + tw_bytes, c->stream>>>(
                    c->frac, c->key_out, c->row_start, c->row_unit, c->density, c->spectrum, c->twiddle, c->heavy_rows, counters, c->light_rows, c->g, mark_rows);
            else
                k_deposit_rows<NFFT, true><<<grid, BLOCK, lds_rows + tw_bytes, c->stream>>>(
                    c->frac, c->key_out, c->row_start, c->row_unit, c->density, c->spectrum, c->twiddle, c->heavy_rows, counters, c->light_rows, c->g, mark_rows);
        }
    }
    if (beside && (e = hipStreamWaitEvent(c->stream, c->ev_join, 0)) != hipSuccess) return e;
    if (planned)
        k_sum_units<NFFT><<<2048, SUM_BLOCK, (size_t)G * sizeof(double) + tw_bytes, c->stream>>>(c->row_start, c->row_unit, c->heavy_rows, counters, c->unit_rows,
                                                                                            c->light_rows, c->density, c->spectrum, c->twiddle, c->g,
                                                                                            next_counters, seen_dev, (uint32_t)(turn + 1));
    return hipGetLastError();
}

// to_spectrum: leave the z-transformed rows in the half spectrum instead of the density mesh (grid sizes of the
// hand-written transforms only)
hipError_t pm_launch_deposit(pm_ctx *c, bool to_spectrum)
{
    if (!to_spectrum) return deposit_dispatch<0>(c);
    switch (c->g.G) {
    case 128: return deposit_dispatch<128>(c);
    case 256: return deposit_dispatch<256>(c);
    case 512: return deposit_dispatch<512>(c);
    case 1024: return deposit_dispatch<1024>(c);
    case 2048: return deposit_dispatch<2048>(c);
    default: return hipErrorInvalidValue;
    }
}

template <typename TP, typename TV>
static void gather_dispatch(pm_ctx *c, const pm_prec<TP> *P, pm_vrec<TV> *const V[2], double dt, bool integrate, float *forces, bool fused)
{
    int64_t n = c->n;
    pm_prec<float> *out = c->pf[c->cur];
    const pm_vrec<TV> *vin = V[c->vcur];
    pm_vrec<TV> *vout = c->v_stale ? V[1 - c->vcur] : V[c->vcur];
    const uint32_t *vmap = c->v_stale ? c->src : nullptr;
    // the drift also marks the particles that changed their cell and estimates how many they are (pm_mover_probe)
    pm_mover_probe probe{c->mover_count, c->movers_seen_dev, 0u, nullptr, 64u};
    if (integrate && c->mover_count && n >= c->rebin_min) {
        // the sample: every 64th workgroup of a large set, more of a small one (about a thousand workgroups where there are as many)
        const int64_t full = n / BLOCK;
        while (probe.every > 1u && full / probe.every < 1024) probe.every >>= 1;
        probe.expect = 4u * (uint32_t)((full + probe.every - 1) / probe.every);
        // the bits are for a binning that starts from this order: not written while the estimates say that far too many
        // particles move for that (a set that calms down is marked from the step after the estimate says so)
        const uint32_t seen = c->movers_seen ? *c->movers_seen : 0u;
        // (nor on a context whose binnings never start from the order before)
        const bool may_rebin = !(c->flags & (PM_FLAG_FULL_SORT | PM_FLAG_NGP | PM_FLAG_F64));
        probe.mask = (may_rebin && (seen == 0xffffffffu || (int64_t)seen * 3 <= n)) ? c->rb_mask : nullptr;
        c->movers_marked = probe.mask != nullptr;
    }
    auto launch = [&](auto mapped) {
        constexpr bool M = decltype(mapped)::value;
        if (c->slab) {
            if constexpr (sizeof(TP) == 4 && sizeof(TV) == 4) {
                const outbox_t ob{c->outbox, c->outbox_key, c->outbox_count, (uint32_t)c->outbox_cap, c->cuts, c->nslabs};
                if (n > PM_STREAM_MIN)
                    k_gather<float, float, true, true, M, true><<<grid_for(n), BLOCK, 0, c->stream>>>(P, out, vin, vout, vmap, nullptr, nullptr, nullptr, c->potential,
                                                                                                     forces, c->key_in, n, c->g, dt, integrate, ob, probe);
                else
                    k_gather<float, float, true, true, M><<<grid_for(n), BLOCK, 0, c->stream>>>(P, out, vin, vout, vmap, nullptr, nullptr, nullptr, c->potential, forces,
                                                                                               c->key_in, n, c->g, dt, integrate, ob, probe);
            }
        } else if (fused) {
            // (the float32 step of a set too large for the caches: the records in and out marked as streaming)
            if constexpr (sizeof(TP) == 4 && sizeof(TV) == 4) {
                if (n > PM_STREAM_MIN) {
                    k_gather<TP, TV, true, false, M, true><<<grid_for(n), BLOCK, 0, c->stream>>>(P, out, vin, vout, vmap, nullptr, nullptr, nullptr, c->potential, forces, c->key_in, n, c->g, dt, integrate, outbox_t{}, probe);
                    return;
                }
            }
            k_gather<TP, TV, true, false, M><<<grid_for(n), BLOCK, 0, c->stream>>>(P, out, vin, vout, vmap, nullptr, nullptr, nullptr, c->potential, forces, c->key_in, n, c->g, dt, integrate, outbox_t{}, probe);
        } else
            k_gather<TP, TV, false, false, M><<<grid_for(n), BLOCK, 0, c->stream>>>(P, out, vin, vout, vmap, c->grad[0], c->grad[1], c->grad[2], nullptr, forces, c->key_in, n, c->g, dt, integrate, outbox_t{}, probe);
    };
    if (vmap) launch(std::true_type{});
    else launch(std::false_type{});
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

hipError_t pm_launch_export_forces64(pm_ctx *c, double *aos)
{
    if (c->n == 0) return hipSuccess;
    const uint32_t *vmap = c->v_stale ? c->src : nullptr;
    if (c->vel_f64) k_export_forces<double, double><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->forces64, c->vd[c->vcur], vmap, aos, c->n);
    else k_export_forces<float, double><<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->forces64, c->vf[c->vcur], vmap, aos, c->n);
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

// ---- slab decomposition ------------------------------------------------------------------------

hipError_t pm_launch_set_ids(pm_ctx *c, const uint32_t *ids)
{
    if (c->n == 0) return hipSuccess;
    k_set_ids<<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->vf[c->vcur], ids, c->n);
    return hipGetLastError();
}

hipError_t pm_launch_pack_outbox(pm_ctx *c, uint32_t count, pm_migrant *dst)
{
    if (count == 0) return hipSuccess;
    k_pack_outbox<<<grid_for(count), BLOCK, 0, c->stream>>>(c->outbox, c->outbox_idx2, dst, count);
    return hipGetLastError();
}

hipError_t pm_launch_outbox_offsets(pm_ctx *c, uint32_t count, uint32_t *offsets)
{
    k_outbox_offsets<<<grid_for(c->nslabs + 1), BLOCK, 0, c->stream>>>(c->outbox_key2, count, c->nslabs, offsets);
    return hipGetLastError();
}

hipError_t pm_launch_append(pm_ctx *c, const pm_migrant *src, int64_t count)
{
    if (count == 0) return hipSuccess;
    k_append<<<grid_for(count), BLOCK, 0, c->stream>>>(src, c->pf[c->cur], c->vf[c->vcur], c->key_in, c->n, count, c->g);
    return hipGetLastError();
}

hipError_t pm_launch_rekey(pm_ctx *c)
{
    if (c->n == 0) return hipSuccess;
    const outbox_t ob{c->outbox, c->outbox_key, c->outbox_count, (uint32_t)c->outbox_cap, c->cuts, c->nslabs};
    k_rekey<<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pf[c->cur], c->vf[c->vcur], c->key_in, c->n, c->g, ob);
    return hipGetLastError();
}

hipError_t pm_launch_plane_counts(pm_ctx *c, unsigned long long *plane_counts)
{
    if (c->n == 0) return hipSuccess;
    const int64_t want = (c->n + BLOCK - 1) / BLOCK;
    k_plane_counts<<<(unsigned)(want < 2048 ? want : 2048), BLOCK, (size_t)c->g.G * sizeof(uint32_t), c->stream>>>(c->pf[c->cur], c->n, c->g, plane_counts);
    return hipGetLastError();
}

hipError_t pm_launch_export_slab(pm_ctx *c, uint32_t *ids, float *pos, float *vel)
{
    if (c->n == 0) return hipSuccess;
    k_export_slab<<<grid_for(c->n), BLOCK, 0, c->stream>>>(c->pf[c->cur], c->vf[c->vcur], c->v_stale ? c->src : nullptr, ids, pos, vel, c->n);
    return hipGetLastError();
}

hipError_t pm_launch_outbox_counts(pm_ctx *c, int64_t *device_counts)
{
    k_outbox_counts<<<grid_for(c->nslabs + 1), BLOCK, 0, c->stream>>>(c->outbox_count, c->nslabs, device_counts);
    return hipGetLastError();
}

#ifdef PM_TRACE
// tuning builds only: copies the wave trace of the last deposit to the host (2 x TRACE_ROWS x 4 words)
extern "C" int pm_debug_trace(void *host, size_t bytes)
{
    if (bytes > sizeof(g_trace)) bytes = sizeof(g_trace);
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), bytes, 0, hipMemcpyDeviceToHost);
}
#endif
