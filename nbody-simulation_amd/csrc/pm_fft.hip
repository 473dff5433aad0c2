// Column FFTs of the Poisson solve for gfx950.
//
// The half spectrum S[i][j][c] (c = 0..G/2, row pitch Ghp complex, 128-byte aligned rows) is
// transformed along j and along i by these kernels, in place.  Both are strided ("column")
// transforms: the transform axis has a large stride and the neighbouring columns are contiguous
// in memory, so a workgroup takes a tile of COLS = 16 neighbouring columns (128 contiguous bytes
// per row) by all N rows, keeps it in LDS (N*16*8 bytes = 64 KiB at N = 512, 128 KiB at N = 1024) and
// runs a Stockham FFT (radix 8, 8, 8 or 16, 8, 8) on the 16 columns side by side.  Every global access of a wave is four
// full 128-byte lines; every LDS access of a 16-lane group is one 128-byte row (conflict free).
//
// The pass along i is fused: forward FFT, Green's-function multiply (functions.py:75-86) with
// the 1/M of the inverse transform (functions.py:89), inverse FFT, one read and one write of the
// spectrum instead of three.  That works without an extra exchange because a thread of the
// Stockham schedule ends the forward transform holding exactly the rows that its first inverse
// stage starts from.
#include "pm_internal.hpp"
#include "pm_fft_dev.hpp"

#include <atomic>
#include <cmath>
#include <type_traits>
#include <vector>

namespace {

using namespace pmfft;

constexpr int COLS = 16;
constexpr int FFT_BLOCK = 256;

// np.fft.fftfreq(G)[a] as float32 (functions.py:75): integer frequency times 1/G in float64, then rounded.
__device__ __forceinline__ float freq(int a, int G, double invG)
{
    int f = (a < (G + 1) / 2) ? a : a - G;
    return (float)((double)f * invG);
}

struct col_pass {
    float2 *spec;
    // element (outer, row, col) of `spec` is at
    //   (outer >> o_shift) * o_hi + (outer & (2^o_shift - 1)) * o_lo + (row >> r_shift) * r_hi + (row & (2^r_shift - 1)) * r_lo + col:
    // plain strides with both shifts 0, the interleaved planes of pm_spec_row otherwise
    int64_t o_hi, o_lo, r_hi, r_lo;
    int o_shift, r_shift;
    int64_t row_stride;     // elements between consecutive rows of `blocks` (BLK != 0)
    int tiles;              // column tiles per outer index
    int ncols;              // valid columns per outer index (G/2 + 1)
    int G;
    float invM;
    float soft;                 // pm_geom::soft
    int outer0;             // mesh index of outer index 0 (Green multiply)
    int tile0;              // first column tile of this launch (a launch may cover a range of tiles only)
    int ntiles;             // column tiles per outer index in this launch (without the ones counted by `lone`)
    int lone;               // workgroups at the start of the launch that take the last tile of C outer indices each
    // BLK != 0 (slab decomposition, transform along y): the other side of the pass is not `spec` but the buffer of an
    // all-to-all, `blocks`, laid out (G / Gs, Gs planes, Gs rows, pitch): row y of plane `outer` lies in block y / Gs
    float2 *blocks;
    int gs_shift;           // log2 Gs
    int64_t block_stride;   // Gs * Gs * pitch
    // forward pass along y of a whole mesh right after a deposit that marked its empty rows (k_plan_units): bit (outer * G +
    // row) says whether the row was written; the others are zeros that are not read.  Null: every row is read.
    const unsigned long long *row_live;
};

// Radix plan of the column transform: N = R1 * 8 * 8 (R1 = 2, 4, 8, 16, 32 at N = 128, 256, 512, 1024, 2048).  A column is
// worked on by TPC = N / EPT threads, each holding EPT of its elements; a block is 16 columns = 16 TPC threads
// (512 or 1024 threads: more waves per CU overlap one tile's loads with another's butterflies better than the 32
// elements per thread that the registers would allow).
// elements of a column per thread, per pass (forward, forward + Green + inverse, inverse) at N = 512; 16 at N = 1024
#ifndef PM_EPT_FWD
#define PM_EPT_FWD 8
#endif
#ifndef PM_EPT_GREEN
#define PM_EPT_GREEN 16
#endif
#ifndef PM_EPT_INV
#define PM_EPT_INV 8
#endif
// columns per workgroup: 16 (128-byte row segments), or 8 for the x pass at N = 1024 (64-byte segments, but half the
// LDS, so that a CU holds two workgroups and one's butterflies run beside the other's loads: -10 % for that pass; the y
// passes and N = 512, whose tiles leave room for two workgroups anyway, are faster with 16)
#ifndef PM_COLS_GREEN_1024
#define PM_COLS_GREEN_1024 8
#endif
template <int N, int DIR, bool GREEN> struct col_plan {
    // (N = 2048 = 32 * 8 * 8: 8 columns are what fits the LDS, 128 KiB; a thread holds the 32 rows of a first-stage butterfly)
    static constexpr int C = N == 2048 ? 8 : (N == 1024 && GREEN) ? PM_COLS_GREEN_1024 : 16;
    static constexpr int EPT = N == 2048 ? 32 : N != 512 ? 16 : GREEN ? PM_EPT_GREEN : DIR < 0 ? PM_EPT_FWD : PM_EPT_INV;
    static constexpr int R1 = N / 64;                   // first radix
    static constexpr int TPC = N / EPT;                 // threads per column
    static constexpr int THREADS = C * TPC;
    static constexpr int NB1 = (N / R1) / TPC;          // butterflies per thread in stage 1
    static constexpr int NB = (N / 8) / TPC;            // ... in stages 2 and 3 (= 4)
    static constexpr size_t LDS_BYTES = (size_t)(N * C + N) * sizeof(float2);
    // waves per SIMD that the registers must leave room for: what the LDS lets a CU hold (two workgroups of 16 waves for the
    // passes along j at N = 512, two of 8 waves for the fused pass and up to N = 1024, one of 8 waves at N = 2048).  The
    // kernels fit without it; the bound is there for the rarely taken LONE path, which may spill rather than cost the
    // whole kernel a workgroup per CU.
    static constexpr int MIN_WAVES = (N == 512 && !GREEN) ? 8 : N <= 1024 ? 4 : 2;
};

// Thread t owns column c = t % C of its workgroup's C columns; jl = t / C is its place among the column's threads.  Stage 1
// (radix R1, no twiddles) takes butterflies j = jl + TPC b and rows j + 64 r; stages 2 and 3 (radix 8)
// take j = jl + TPC b and rows j + (N/8) r.  After stage 3 the thread holds the rows jl + TPC m, m < EPT:
// exactly the inputs of its stage-1 butterflies, which is what lets the Green pass turn round in registers.
// BLK = 1: the results go to p.blocks instead of back to p.spec; BLK = 2: the inputs come from p.blocks; BLK = 3 (Green
// pass only): the multiply includes the Gaussian softening factor exp(-p.soft k2) -- an instantiation of its own, so that
// the pass without softening stays the code it was (as a run-time branch the factor cost that pass 6 %).
//
// With G/2 + 1 = 16 q + 1 columns the last tile of every outer index holds one column, G/2, and fifteen that do not exist.
// Those tiles are taken C outer indices at a time instead (LONE: the p.lone first workgroups of the launch; thread column
// c works on column G/2 of outer index C * n + c), which saves one tile in q + 1 for the same memory accesses.
template <int N, int DIR, bool GREEN, int BLK, bool LONE>
__device__ __forceinline__ void col_tile(const col_pass &p, const float2 *__restrict__ twiddle, float2 *lds, int lb)
{
    using plan = col_plan<N, DIR, GREEN>;
    constexpr int C = plan::C, R1 = plan::R1, TPC = plan::TPC, NB1 = plan::NB1, NB = plan::NB, Q = N / 8;
    float2 *tile = lds;                             // [N][C] tile, then the N twiddles
    float2 *tw = lds + N * C;
    const int t = threadIdx.x, c = t & (C - 1), jl = t / C;
    constexpr int S = COLS / C;
    const int per_outer = p.ntiles * S;
    const int outer = LONE ? lb * C + c : lb / per_outer;
    const int col = LONE ? p.ncols - 1 : p.tile0 * COLS + (lb % per_outer) * C + c;
    const bool valid = col < p.ncols;
    float2 *g = p.spec + (int64_t)(outer >> p.o_shift) * p.o_hi + (int64_t)(outer & ((1 << p.o_shift) - 1)) * p.o_lo + col;
    auto in_spec = [&](int row) -> float2 & {
        return g[(int64_t)(row >> p.r_shift) * p.r_hi + (int64_t)(row & ((1 << p.r_shift) - 1)) * p.r_lo];
    };
    // the same column in the all-to-all buffer: plane `outer`'s rows of block b start at (b * Gs + outer) * Gs * pitch
    float2 *gb = (BLK == 1 || BLK == 2) ? p.blocks + ((int64_t)outer << p.gs_shift) * p.row_stride + col : nullptr;
    auto in_blocks = [&](int row) -> float2 & {
        return gb[(int64_t)(row >> p.gs_shift) * p.block_stride + (int64_t)(row & ((1 << p.gs_shift) - 1)) * p.row_stride];
    };

    for (int q = t; q < N; q += plan::THREADS) tw[q] = twiddle[q];

    constexpr int EPT = plan::EPT;
    float2 v[EPT];
    // ---- stage 1 inputs from global: v[b * R1 + r] = row jl + TPC b + 64 r
    // (rows jl + 64 m, m = 0 .. N / 64 - 1, of plane `outer`: bit jl of each of the plane's N / 64 words of row_live)
    constexpr bool MARKED = BLK == 0 && !GREEN && DIR < 0 && TPC == 64 && NB1 == 1;
    const unsigned long long *live = MARKED ? p.row_live : nullptr;
#pragma unroll
    for (int b = 0; b < NB1; ++b)
#pragma unroll
        for (int r = 0; r < R1; ++r) {
            int row = jl + TPC * b + 64 * r;
            if (BLK == 2) v[b * R1 + r] = valid ? in_blocks(row) : make_float2(0.0f, 0.0f);
            else if (MARKED && live != nullptr) {
                // (the word is the same for all threads of the workgroup: a scalar load)
                const unsigned long long w = live[(size_t)outer * (N / 64) + (b * R1 + r)];
                v[b * R1 + r] = (valid && ((w >> jl) & 1ull)) ? in_spec(row) : make_float2(0.0f, 0.0f);
            } else v[b * R1 + r] = valid ? in_spec(row) : make_float2(0.0f, 0.0f);
        }

    auto stage1_out = [&](auto dir) {
        constexpr int D = decltype(dir)::value;
#pragma unroll
        for (int b = 0; b < NB1; ++b) {
            fft_r<R1, D>(v + b * R1);
            int j = jl + TPC * b;
#pragma unroll
            for (int r = 0; r < R1; ++r) tile[(R1 * j + r) * C + c] = v[b * R1 + r];
        }
    };
    auto stage2 = [&](auto dir) {
        constexpr int D = decltype(dir)::value;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[b * 8 + r] = tile[(jl + TPC * b + Q * r) * C + c];
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int j = jl + TPC * b, k = j & (R1 - 1);
#pragma unroll
            for (int r = 1; r < 8; ++r) v[b * 8 + r] = cmul(v[b * 8 + r], twid<D>(tw[8 * r * k]));       // N / (R1 * 8) = 8
            fft8<D>(v + b * 8);
            int j0 = (j / R1) * (R1 * 8) + k;
#pragma unroll
            for (int r = 0; r < 8; ++r) tile[(j0 + R1 * r) * C + c] = v[b * 8 + r];
        }
    };
    auto stage3 = [&](auto dir) {
        constexpr int D = decltype(dir)::value;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[b * 8 + r] = tile[(jl + TPC * b + Q * r) * C + c];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int j = jl + TPC * b;
#pragma unroll
            for (int r = 1; r < 8; ++r) v[b * 8 + r] = cmul(v[b * 8 + r], twid<D>(tw[r * j]));
            fft8<D>(v + b * 8);          // v[b * 8 + r] is now element j + (N/8) r of the transform
        }
    };
    using dir_t = std::integral_constant<int, DIR>;
    using inv_t = std::integral_constant<int, 1>;

    __syncthreads();                // twiddles in place
    stage1_out(dir_t{});
    __syncthreads();
    stage2(dir_t{});
    __syncthreads();
    stage3(dir_t{});

    if (GREEN) {
        // rows are frequencies along x; outer is the y index, col the z index (functions.py:75-86)
        const int G = p.G;
        const double invG = 1.0 / (double)G;
        const int yy = outer + p.outer0;
        const float ky = freq(yy, G, invG), kz = freq(col, G, invG);
        const float kyz = ky * ky;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int a = jl + TPC * b + Q * r;
                float kx = freq(a, G, invG);
                float k2 = __fadd_rn(__fadd_rn(__fmul_rn(kx, kx), kyz), __fmul_rn(kz, kz));
                float scl = ((a | yy | col) == 0) ? 0.0f : __fmul_rn(__frcp_rn(k2), p.invM);
                if constexpr (BLK == 3) scl *= expf(-p.soft * k2);
                v[b * 8 + r].x *= scl; v[b * 8 + r].y *= scl;
            }
        if constexpr (R1 != 8) {
            // from "row jl + TPC (b + 4 r)" to the stage-1 order "row jl + TPC (b' + NB1 r')"
            float2 u[EPT];
#pragma unroll
            for (int m = 0; m < EPT; ++m) u[(m % NB1) * R1 + m / NB1] = v[(m % NB) * 8 + m / NB];
#pragma unroll
            for (int m = 0; m < EPT; ++m) v[m] = u[m];
        }
        __syncthreads();            // every thread has read its stage-3 inputs
        stage1_out(inv_t{});
        __syncthreads();
        stage2(inv_t{});
        __syncthreads();
        stage3(inv_t{});
    }

    if (valid) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int row = jl + TPC * b + Q * r;
                if (BLK == 1) in_blocks(row) = v[b * 8 + r];
                else in_spec(row) = v[b * 8 + r];
            }
    }
}

template <int N, int DIR, bool GREEN, int BLK = 0>
__global__ void __launch_bounds__((col_plan<N, DIR, GREEN>::THREADS), (col_plan<N, DIR, GREEN>::MIN_WAVES))
k_col_fft(col_pass p, const float2 *__restrict__ twiddle)
{
    extern __shared__ float2 lds[];
    // S = 16 / C workgroups share a 16-column tile.  The hardware hands workgroup b to XCD b % 8: the two halves of a
    // tile are given to workgroups b and b + 8, so that the 128-byte lines they share are fetched into one L2.
    constexpr int S = COLS / col_plan<N, DIR, GREEN>::C;
    // (the lone tiles first: at the end of the launch they would be a round of their own)
    int lb = blockIdx.x;
    if (lb < p.lone) {
        col_tile<N, DIR, GREEN, BLK, true>(p, twiddle, lds, lb);
        return;
    }
    lb -= p.lone;
    const int full = (int)gridDim.x - p.lone;
    if (S == 2 && (lb | 15) < full) lb = (lb & ~15) + 2 * (lb & 7) + ((lb >> 3) & 1);
    col_tile<N, DIR, GREEN, BLK, false>(p, twiddle, lds, lb);
}

// ---- z axis: real <-> half-complex transforms of contiguous rows ----------------------------------
//
// One wave per mesh row.  A row of N reals is read as M = N/2 complex numbers z[n] = x[2n] + i x[2n+1],
// transformed by an M-point Stockham FFT (one butterfly per lane per stage, lane j always reading
// elements j + 64 r; radix 4 at N = 512, radix 8 at N = 1024), and split into the M + 1 values of the
// half spectrum:
//   X[k] = (Z[k] + conj Z[M-k])/2 - (i/2) w^k (Z[k] - conj Z[M-k]),   w = exp(-2 pi i / N).
// The inverse runs the same steps backwards:
//   Z[k] = (X[k] + conj X[M-k]) + i w^-k (X[k] - conj X[M-k]),  z = unnormalised inverse FFT of Z.
// Every global access of a wave is a contiguous 512-byte run.

#ifndef PM_ROWS_PER_WAVE
#define PM_ROWS_PER_WAVE 1
#endif
constexpr int ROWS_PER_WAVE = PM_ROWS_PER_WAVE;

template <int N>
__global__ void __launch_bounds__(FFT_BLOCK)
k_row_r2c(const float *__restrict__ in, float2 *__restrict__ out, int pitch, const float2 *__restrict__ twiddle, int sb)
{
    constexpr int E = N / 128, M = N / 2;
    __shared__ float2 tw[N];
    __shared__ float2 bufs[FFT_BLOCK / 64][M];
    const int j = threadIdx.x & 63, w = threadIdx.x >> 6;
    float2 *buf = bufs[w];
    for (int q = threadIdx.x; q < N; q += FFT_BLOCK) tw[q] = twiddle[q];
    __syncthreads();
    const int64_t row0 = ((int64_t)blockIdx.x * (FFT_BLOCK / 64) + w) * ROWS_PER_WAVE;
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const float2 *src = (const float2 *)(in + (row0 + rr) * N);
        float2 v[E];
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = src[j + 64 * r];
        wave_fft<E, -1>(v, buf, tw, j);
        const int64_t row = row0 + rr;          // mesh row i * N + j
        wave_r2c_split<E>(v, buf, tw, j, out + pm_spec_row(N, sb, (int)(row / N), (int)(row % N)) * pitch);
    }
}

template <int N>
__global__ void __launch_bounds__(FFT_BLOCK)
k_row_c2r(const float2 *__restrict__ in, float *__restrict__ out, int pitch, const float2 *__restrict__ twiddle, int sb,
          const unsigned long long *__restrict__ need)
{
    constexpr int E = N / 128, M = N / 2;
    __shared__ float2 tw[N];
    __shared__ float2 bufs[FFT_BLOCK / 64][M + 4];
    const int j = threadIdx.x & 63, w = threadIdx.x >> 6;
    float2 *buf = bufs[w];
    // last planes first: the pass along y before this one ends with them, and the gather after it starts with the first
    const int64_t row0 = ((int64_t)(gridDim.x - 1 - blockIdx.x) * (FFT_BLOCK / 64) + w) * ROWS_PER_WAVE;
    // need: a bit per row, set where the row will be read; the others are left as they are.  (One word per wave, asked for
    // together with the twiddles: a wave is a short chain of waits and should not start with one more.)
    static_assert(64 % ROWS_PER_WAVE == 0, "a wave's rows share a word of the marks");
    const unsigned long long needed = need != nullptr ? need[row0 >> 6] : ~0ull;
    for (int q = threadIdx.x; q < N; q += FFT_BLOCK) tw[q] = twiddle[q];
    __syncthreads();
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const int64_t row = row0 + rr;
        if (!((needed >> (row & 63)) & 1ull)) continue;
        const float2 *src = in + pm_spec_row(N, sb, (int)(row / N), (int)(row % N)) * pitch;
        float2 x[E], v[E];
#pragma unroll
        for (int r = 0; r < E; ++r) x[r] = src[j + 64 * r];
        const float2 xm = (j == 0) ? src[M] : make_float2(0.0f, 0.0f);
        wave_c2r_merge<E>(x, xm, v, buf, tw, j);
        wave_fft<E, 1>(v, buf, tw, j);
        float2 *dst = (float2 *)(out + (row0 + rr) * N);
#pragma unroll
        for (int r = 0; r < E; ++r) dst[j + 64 * r] = v[r];
    }
}

template <int N, int DIR, bool GREEN, int BLK = 0>
hipError_t launch_col(pm_ctx *c, const col_pass &p, int blocks)
{
    if (col_plan<N, DIR, GREEN>::LDS_BYTES > 64 * 1024) {
        // more LDS than a kernel gets by default: allowed once per kernel and device (a bit per device; two threads
        // that both find it unset both set it, which is harmless)
        static std::atomic<uint64_t> allowed{0};
        const uint64_t bit = 1ull << (c->device & 63);
        if (!(allowed.load(std::memory_order_acquire) & bit)) {
            hipError_t e = hipFuncSetAttribute((const void *)k_col_fft<N, DIR, GREEN, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)col_plan<N, DIR, GREEN>::LDS_BYTES);
            if (e != hipSuccess) return e;
            allowed.fetch_or(bit, std::memory_order_release);
        }
    }
    constexpr int C = col_plan<N, DIR, GREEN>::C;
    col_pass q = p;
    const int outers = p.ntiles > 0 ? blocks / p.ntiles : 0;
#ifndef PM_COL_NO_LONE
    if (p.ncols % COLS == 1 && p.tile0 + p.ntiles == p.tiles && outers > 0 && outers % C == 0) { q.lone = outers / C; q.ntiles -= 1; }
#endif
    const int grid = outers * q.ntiles * (COLS / C) + q.lone;
    if (grid == 0) return hipSuccess;
    k_col_fft<N, DIR, GREEN, BLK><<<grid, col_plan<N, DIR, GREEN>::THREADS, col_plan<N, DIR, GREEN>::LDS_BYTES, c->stream>>>(q, c->twiddle);
    return hipGetLastError();
}

// f(std::integral_constant<int, G>) for the grid sizes the hand-written transforms are built for
template <typename F> hipError_t with_grid(int G, F f)
{
    switch (G) {
    case 128: return f(std::integral_constant<int, 128>{});
    case 256: return f(std::integral_constant<int, 256>{});
    case 512: return f(std::integral_constant<int, 512>{});
    case 1024: return f(std::integral_constant<int, 1024>{});
    case 2048: return f(std::integral_constant<int, 2048>{});
    default: return hipErrorInvalidValue;
    }
}

// dir < 0 forward, > 0 inverse, 0: forward, Green multiply, inverse
hipError_t launch_col_any(pm_ctx *c, int dir, const col_pass &p, int blocks)
{
    return with_grid(c->g.G, [&](auto n) {
        constexpr int N = decltype(n)::value;
        if (dir == 0) return p.soft != 0.0f ? launch_col<N, -1, true, 3>(c, p, blocks) : launch_col<N, -1, true>(c, p, blocks);
        return dir < 0 ? launch_col<N, -1, false>(c, p, blocks) : launch_col<N, 1, false>(c, p, blocks);
    });
}

// sb: the interleaving of `spec` (0 for the plane buffers of a slab)
hipError_t launch_rows(pm_ctx *c, bool forward, const float *real_in, float *real_out, float2 *spec, int64_t rows, int sb,
                       const unsigned long long *need = nullptr)
{
    const int blocks = (int)(rows / ((FFT_BLOCK / 64) * ROWS_PER_WAVE));
    return with_grid(c->g.G, [&](auto n) {
        constexpr int N = decltype(n)::value;
        if (forward) k_row_r2c<N><<<blocks, FFT_BLOCK, 0, c->stream>>>(real_in, spec, c->g.Ghp, c->twiddle, sb);
        else k_row_c2r<N><<<blocks, FFT_BLOCK, 0, c->stream>>>(spec, real_out, c->g.Ghp, c->twiddle, sb, need);
        return hipGetLastError();
    });
}

}  // namespace

bool pm_fft_custom_supported(int G) { return G == 128 || G == 256 || G == 512 || G == 1024 || G == 2048; }

// log2 of the planes whose rows the whole-mesh half spectrum interleaves (pm_spec_row).  At N = 1024 a third of the
// address translations of the pass along i miss the CU's TLB in plain order (1024 rows, 4.3 MB apart, per tile); four
// interleaved planes cut its time by 20 % and cost the passes along j, whose rows move four times further apart,
// 4 %: -0.45 ms per step.  More planes help the one and hurt the other more; at N = 512 the plain order is fastest.
#ifndef PM_SPEC_B_512
#define PM_SPEC_B_512 0
#endif
#ifndef PM_SPEC_B_1024
#define PM_SPEC_B_1024 2
#endif
int pm_fft_spectrum_interleave(int G) { return G == 512 ? PM_SPEC_B_512 : G == 1024 ? PM_SPEC_B_1024 : 0; }

int pm_fft_pitch(int G)
{
    int gh = G / 2 + 1;
    return pm_fft_custom_supported(G) ? (gh + COLS - 1) / COLS * COLS : gh;
}

// exp(-2 pi i q / G), q = 0..G-1, rounded from float64
hipError_t pm_fft_upload_twiddles(pm_ctx *c)
{
    const int G = c->g.G;
    std::vector<float2> w(G);
    for (int q = 0; q < G; ++q) {
        double a = -2.0 * M_PI * (double)q / (double)G;
        w[q] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    hipError_t e = hipMalloc((void **)&c->twiddle, G * sizeof(float2));
    if (e != hipSuccess) return e;
    return hipMemcpy(c->twiddle, w.data(), G * sizeof(float2), hipMemcpyHostToDevice);
}

// z axis: density (G,G,G) -> half spectrum (forward), half spectrum -> potential (inverse)
hipError_t pm_launch_row_fft(pm_ctx *c, bool forward, bool needed_only)
{
    return launch_rows(c, forward, c->density, c->potential, c->spectrum, (int64_t)c->g.G * c->g.G, c->g.sb,
                       !forward && needed_only ? c->row_need : nullptr);
}

static col_pass pass_for(pm_ctx *c, float2 *spec)
{
    const int G = c->g.G, P = c->g.Ghp;
    col_pass p{};
    p.spec = spec; p.G = G; p.ncols = G / 2 + 1; p.tiles = P / COLS; p.tile0 = 0; p.ntiles = p.tiles;
    p.invM = (float)(1.0 / (double)c->g.M);
    p.soft = c->g.soft;
    return p;
}

static void plain_strides(col_pass &p, int64_t outer_stride, int64_t row_stride)
{
    p.o_shift = p.r_shift = 0;
    p.o_hi = outer_stride; p.r_hi = row_stride; p.o_lo = p.r_lo = 0;
}

// the context's own half spectrum (rows interleaved as pm_spec_row says): along j (outer = plane i) or along i (outer = j)
static void spectrum_strides(pm_ctx *c, col_pass &p, bool along_i)
{
    const int64_t G = c->g.G, P = c->g.Ghp;
    const int sb = c->g.sb;
    if (along_i) { p.o_shift = 0; p.o_hi = P << sb; p.o_lo = 0; p.r_shift = sb; p.r_hi = (G << sb) * P; p.r_lo = P; }
    else { p.o_shift = sb; p.o_hi = (G << sb) * P; p.o_lo = P; p.r_shift = 0; p.r_hi = P << sb; p.r_lo = 0; }
}

// dir < 0: forward along j.  dir > 0: inverse along j.  dir == 0: forward along i, Green multiply, inverse along i.
hipError_t pm_launch_col_fft(pm_ctx *c, int dir, bool rows_marked)
{
    const int G = c->g.G;
    col_pass p = pass_for(c, c->spectrum);
    spectrum_strides(c, p, dir == 0);
    if (rows_marked && dir < 0) p.row_live = c->row_live;
    return launch_col_any(c, dir, p, G * p.tiles);
}

// The same on column tiles [tile0, tile0 + ntiles) only (a tile is 16 neighbouring columns c of all rows).
hipError_t pm_launch_col_fft_tiles(pm_ctx *c, int dir, int tile0, int ntiles, bool rows_marked)
{
    const int G = c->g.G;
    col_pass p = pass_for(c, c->spectrum);
    spectrum_strides(c, p, dir == 0);
    p.tile0 = tile0; p.ntiles = ntiles;
    if (rows_marked && dir < 0) p.row_live = c->row_live;
    return launch_col_any(c, dir, p, G * ntiles);
}

int pm_fft_tiles(const pm_ctx *c) { return c->g.Ghp / COLS; }

// ---- the same passes on some planes of the mesh (slab decomposition) ------------------------------

hipError_t pm_launch_row_fft_planes(pm_ctx *c, bool forward, const float *real_in, float *real_out, float2 *spec, int planes)
{
    return launch_rows(c, forward, real_in, real_out, spec, (int64_t)planes * c->g.G, 0);
}

hipError_t pm_launch_col_fft_planes(pm_ctx *c, int dir, float2 *spec, int planes)
{
    const int G = c->g.G, P = c->g.Ghp;
    col_pass p = pass_for(c, spec);
    plain_strides(p, (int64_t)G * P, P);
    return launch_col_any(c, dir < 0 ? -1 : 1, p, planes * p.tiles);
}

// The pass along y of the slab's `planes` planes with the all-to-all buffer on one side: forward from spec into blocks,
// or inverse from blocks into spec -- for planes [first, first + count) of them.  planes must be a power of two.
hipError_t pm_launch_col_fft_blocks(pm_ctx *c, int dir, float2 *spec, float2 *blocks, int planes, int first, int count)
{
    const int G = c->g.G, P = c->g.Ghp;
    col_pass p = pass_for(c, spec + (int64_t)first * G * P);
    plain_strides(p, (int64_t)G * P, P);
    p.row_stride = P;
    p.gs_shift = 0;
    while ((1 << p.gs_shift) < planes) ++p.gs_shift;
    p.blocks = blocks + ((int64_t)first << p.gs_shift) * P;     // plane `first`'s rows of block 0
    p.block_stride = (int64_t)planes * planes * P;
    const int nb = count * p.tiles;
    if (nb == 0) return hipSuccess;
    return with_grid(G, [&](auto n) {
        constexpr int N = decltype(n)::value;
        return dir < 0 ? launch_col<N, -1, false, 1>(c, p, nb) : launch_col<N, 1, false, 2>(c, p, nb);
    });
}

// spec_t: (G, ny, Ghp), transform axis first; its rows are mesh rows y0 .. y0 + ny - 1
hipError_t pm_launch_solve_x(pm_ctx *c, float2 *spec_t, int y0, int ny)
{
    const int P = c->g.Ghp;
    col_pass p = pass_for(c, spec_t);
    plain_strides(p, P, (int64_t)ny * P);
    p.outer0 = y0;
    return launch_col_any(c, 0, p, ny * p.tiles);
}
