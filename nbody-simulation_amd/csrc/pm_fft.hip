// Column FFTs of the Poisson solve for gfx950.
//
// The half spectrum S[i][j][c] (c = 0..G/2, row pitch Ghp complex, 128-byte aligned rows) is
// transformed along j and along i by these kernels, in place.  Both are strided ("column")
// transforms: the transform axis has a large stride and the neighbouring columns are contiguous
// in memory, so a workgroup takes a tile of COLS = 16 neighbouring columns (128 contiguous bytes
// per row) by all N rows, keeps it in LDS (N*16*8 bytes = 64 KiB at N = 512) and runs a
// radix-8 Stockham FFT on the 16 columns side by side.  Every global access of a wave is four
// full 128-byte lines; every LDS access of a 16-lane group is one 128-byte row (conflict free).
//
// The pass along i is fused: forward FFT, Green's-function multiply (functions.py:75-86) with
// the 1/M of the inverse transform (functions.py:89), inverse FFT, one read and one write of the
// spectrum instead of three.  That works without an extra exchange because a thread of the
// Stockham schedule ends the forward transform holding exactly the rows (n + 64 r) that the
// first inverse stage starts from.
#include "pm_internal.hpp"

#include <cmath>
#include <vector>

namespace {

constexpr int COLS = 16;
constexpr int FFT_BLOCK = 256;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// multiply by -i (forward) or +i (inverse)
template <int DIR> __device__ __forceinline__ float2 rot90(float2 a)
{
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}
template <int DIR> __device__ __forceinline__ float2 twid(float2 w) { return DIR < 0 ? w : make_float2(w.x, -w.y); }

// 8-point DFT, in place, natural order in and out.  DIR = -1 forward (exp(-2 pi i nk/8)), +1 inverse.
template <int DIR> __device__ __forceinline__ void fft8(float2 *v)
{
    const float h = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    float2 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    float2 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    float2 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    // a5 *= w8, a6 *= w8^2, a7 *= w8^3 with w8 = exp(DIR * i pi/4)
    a5 = DIR < 0 ? make_float2(h * (a5.x + a5.y), h * (a5.y - a5.x)) : make_float2(h * (a5.x - a5.y), h * (a5.y + a5.x));
    a6 = rot90<DIR>(a6);
    a7 = DIR < 0 ? make_float2(h * (a7.y - a7.x), -h * (a7.x + a7.y)) : make_float2(-h * (a7.x + a7.y), h * (a7.x - a7.y));
    float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = rot90<DIR>(csub(a1, a3));
    float2 b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = rot90<DIR>(csub(a5, a7));
    v[0] = cadd(b0, b1); v[4] = csub(b0, b1); v[2] = cadd(b2, b3); v[6] = csub(b2, b3);
    v[1] = cadd(b4, b5); v[5] = csub(b4, b5); v[3] = cadd(b6, b7); v[7] = csub(b6, b7);
}

// np.fft.fftfreq(G)[a] as float32 (functions.py:75): integer frequency times 1/G in float64, then rounded.
__device__ __forceinline__ float freq(int a, int G, double invG)
{
    int f = (a < (G + 1) / 2) ? a : a - G;
    return (float)((double)f * invG);
}

struct col_pass {
    float2 *spec;
    int64_t outer_stride;   // elements between consecutive `outer` indices
    int64_t row_stride;     // elements between consecutive rows of the transform axis
    int tiles;              // column tiles per outer index
    int ncols;              // valid columns per outer index (G/2 + 1)
    int G;
    float invM;
};

// N = 512 = 8*8*8.  Thread t owns column c = t & 15 and the butterflies j = (t >> 4) + 16 b, b = 0..3.
// Every stage reads rows j + 64 r (r = 0..7); stage s writes rows expand(j) + r * 8^s.
template <int DIR, bool GREEN>
__global__ void __launch_bounds__(FFT_BLOCK, 2)
k_col_fft512(col_pass p, const float2 *__restrict__ twiddle)
{
    constexpr int N = 512, NB = 4;
    extern __shared__ float2 lds[];                 // [N][COLS] tile, then the N twiddles
    float2 *tile = lds;
    float2 *tw = lds + N * COLS;
    const int t = threadIdx.x, c = t & (COLS - 1), jl = t >> 4;
    const int outer = blockIdx.x / p.tiles, ct = blockIdx.x % p.tiles;
    const int col = ct * COLS + c;
    const bool valid = col < p.ncols;
    float2 *g = p.spec + (int64_t)outer * p.outer_stride + col;

    for (int q = t; q < N; q += FFT_BLOCK) tw[q] = twiddle[q];

    float2 v[NB][8];
    // ---- stage 1 (Ns = 1): from global, no twiddles
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int row = jl + 16 * b + 64 * r;
            v[b][r] = valid ? g[(int64_t)row * p.row_stride] : make_float2(0.0f, 0.0f);
        }

    auto stage1_out = [&](auto dir) {
        constexpr int D = decltype(dir)::value;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            fft8<D>(v[b]);
            int j = jl + 16 * b;
#pragma unroll
            for (int r = 0; r < 8; ++r) tile[(8 * j + r) * COLS + c] = v[b][r];
        }
    };
    auto stage2 = [&](auto dir) {
        constexpr int D = decltype(dir)::value;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[b][r] = tile[(jl + 16 * b + 64 * r) * COLS + c];
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int j = jl + 16 * b, k = j & 7;
#pragma unroll
            for (int r = 1; r < 8; ++r) v[b][r] = cmul(v[b][r], twid<D>(tw[8 * r * k]));
            fft8<D>(v[b]);
            int j0 = (j >> 3) * 64 + k;
#pragma unroll
            for (int r = 0; r < 8; ++r) tile[(j0 + 8 * r) * COLS + c] = v[b][r];
        }
    };
    auto stage3 = [&](auto dir) {
        constexpr int D = decltype(dir)::value;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[b][r] = tile[(jl + 16 * b + 64 * r) * COLS + c];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int j = jl + 16 * b;
#pragma unroll
            for (int r = 1; r < 8; ++r) v[b][r] = cmul(v[b][r], twid<D>(tw[r * j]));
            fft8<D>(v[b]);          // v[b][r] is now element j + 64 r of the transform
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
        const float ky = freq(outer, G, invG), kz = freq(col, G, invG);
        const float kyz = ky * ky;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int a = jl + 16 * b + 64 * r;
                float kx = freq(a, G, invG);
                float k2 = __fadd_rn(__fadd_rn(__fmul_rn(kx, kx), kyz), __fmul_rn(kz, kz));
                float scl = ((a | outer | col) == 0) ? 0.0f : __fmul_rn(__frcp_rn(k2), p.invM);
                v[b][r].x *= scl; v[b][r].y *= scl;
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
                int row = jl + 16 * b + 64 * r;
                g[(int64_t)row * p.row_stride] = v[b][r];
            }
    }
}

constexpr size_t LDS_BYTES_512 = (512 * COLS + 512) * sizeof(float2);

template <int DIR, bool GREEN>
hipError_t launch512(pm_ctx *c, const col_pass &p, int blocks)
{
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void *)k_col_fft512<DIR, GREEN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES_512);
        if (e != hipSuccess) return e;
        attr = true;
    }
    k_col_fft512<DIR, GREEN><<<blocks, FFT_BLOCK, LDS_BYTES_512, c->stream>>>(p, c->twiddle);
    return hipGetLastError();
}

}  // namespace

bool pm_fft_custom_supported(int G) { return G == 512; }

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

// dir < 0: forward along j.  dir > 0: inverse along j.  dir == 0: forward along i, Green multiply, inverse along i.
hipError_t pm_launch_col_fft(pm_ctx *c, int dir)
{
    const int G = c->g.G, P = c->g.Ghp;
    col_pass p{};
    p.spec = c->spectrum; p.G = G; p.ncols = G / 2 + 1; p.tiles = P / COLS;
    p.invM = (float)(1.0 / (double)c->g.M);
    if (dir == 0) {
        p.outer_stride = P; p.row_stride = (int64_t)G * P;
        return launch512<-1, true>(c, p, G * p.tiles);
    }
    p.outer_stride = (int64_t)G * P; p.row_stride = P;
    if (dir < 0) return launch512<-1, false>(c, p, G * p.tiles);
    return launch512<1, false>(c, p, G * p.tiles);
}
