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
#include "pm_fft_dev.hpp"

#include <cmath>
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
    int64_t outer_stride;   // elements between consecutive `outer` indices
    int64_t row_stride;     // elements between consecutive rows of the transform axis
    int tiles;              // column tiles per outer index
    int ncols;              // valid columns per outer index (G/2 + 1)
    int G;
    float invM;
    int outer0;             // mesh index of outer index 0 (Green multiply)
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
        const int yy = outer + p.outer0;
        const float ky = freq(yy, G, invG), kz = freq(col, G, invG);
        const float kyz = ky * ky;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int a = jl + 16 * b + 64 * r;
                float kx = freq(a, G, invG);
                float k2 = __fadd_rn(__fadd_rn(__fmul_rn(kx, kx), kyz), __fmul_rn(kz, kz));
                float scl = ((a | yy | col) == 0) ? 0.0f : __fmul_rn(__frcp_rn(k2), p.invM);
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

// ---- z axis: real <-> half-complex transforms of contiguous rows ----------------------------------
//
// One wave per mesh row.  A row of N = 512 reals is read as 256 complex numbers z[n] = x[2n] + i x[2n+1],
// transformed by a 256-point radix-4 Stockham FFT (one butterfly per lane per stage, lane j always
// reading elements j + 64 r), and split into the 257 values of the half spectrum:
//   X[k] = (Z[k] + conj Z[256-k])/2 - (i/2) w^k (Z[k] - conj Z[256-k]),   w = exp(-2 pi i / 512).
// The inverse runs the same steps backwards:
//   Z[k] = (X[k] + conj X[256-k]) + i w^-k (X[k] - conj X[256-k]),  z = unnormalised inverse FFT of Z.
// Every global access of a wave is a contiguous 512-byte run.

constexpr int ROWS_PER_WAVE = 4;

__global__ void __launch_bounds__(FFT_BLOCK)
k_row_r2c512(const float *__restrict__ in, float2 *__restrict__ out, int pitch, const float2 *__restrict__ twiddle)
{
    __shared__ float2 tw[512];
    __shared__ float2 bufs[FFT_BLOCK / 64][256];
    const int j = threadIdx.x & 63, w = threadIdx.x >> 6;
    float2 *buf = bufs[w];
    for (int q = threadIdx.x; q < 512; q += FFT_BLOCK) tw[q] = twiddle[q];
    __syncthreads();
    const int64_t row0 = ((int64_t)blockIdx.x * (FFT_BLOCK / 64) + w) * ROWS_PER_WAVE;
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const float2 *src = (const float2 *)(in + (row0 + rr) * 512);
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = src[j + 64 * r];
        wave_fft256<-1>(v, buf, tw, j);
        wave_r2c_split512(v, buf, tw, j, out + (row0 + rr) * pitch);
    }
}

__global__ void __launch_bounds__(FFT_BLOCK)
k_row_c2r512(const float2 *__restrict__ in, float *__restrict__ out, int pitch, const float2 *__restrict__ twiddle)
{
    __shared__ float2 tw[512];
    __shared__ float2 bufs[FFT_BLOCK / 64][260];
    const int j = threadIdx.x & 63, w = threadIdx.x >> 6;
    float2 *buf = bufs[w];
    for (int q = threadIdx.x; q < 512; q += FFT_BLOCK) tw[q] = twiddle[q];
    __syncthreads();
    const int64_t row0 = ((int64_t)blockIdx.x * (FFT_BLOCK / 64) + w) * ROWS_PER_WAVE;
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const float2 *src = in + (row0 + rr) * pitch;
        float2 x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = src[j + 64 * r];
        float2 xn = (j == 0) ? src[256] : make_float2(0.0f, 0.0f);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[j + 64 * r] = x[r];
        if (j == 0) buf[256] = xn;
        wave_sync();
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = j + 64 * r;
            float2 xk = x[r], xp = buf[256 - k];
            if (k == 0) { xk.y = 0.0f; xp.y = 0.0f; }                      // X[0] and X[256] of a real row are real
            const float2 e = make_float2(xk.x + xp.x, xk.y - xp.y);        // X[k] + conj X[256-k]
            const float2 d = make_float2(xk.x - xp.x, xk.y + xp.y);        // X[k] - conj X[256-k]
            const float2 wk = tw[k];
            const float2 t = cmul(make_float2(wk.x, -wk.y), d);            // w^-k d
            v[r] = make_float2(e.x - t.y, e.y + t.x);                      // e + i t
        }
        wave_fft256<1>(v, buf, tw, j);
        float2 *dst = (float2 *)(out + (row0 + rr) * 512);
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[j + 64 * r] = v[r];
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

// z axis: density (G,G,G) -> half spectrum (forward), half spectrum -> potential (inverse)
hipError_t pm_launch_row_fft(pm_ctx *c, bool forward)
{
    const int G = c->g.G;
    const int blocks = (int)((int64_t)G * G / ((FFT_BLOCK / 64) * ROWS_PER_WAVE));
    if (forward) k_row_r2c512<<<blocks, FFT_BLOCK, 0, c->stream>>>(c->density, c->spectrum, c->g.Ghp, c->twiddle);
    else k_row_c2r512<<<blocks, FFT_BLOCK, 0, c->stream>>>(c->spectrum, c->potential, c->g.Ghp, c->twiddle);
    return hipGetLastError();
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

// ---- the same passes on some planes of the mesh (slab decomposition) ------------------------------

hipError_t pm_launch_row_fft_planes(pm_ctx *c, bool forward, const float *real_in, float *real_out, float2 *spec, int planes)
{
    const int G = c->g.G;
    const int blocks = (int)((int64_t)planes * G / ((FFT_BLOCK / 64) * ROWS_PER_WAVE));
    if (forward) k_row_r2c512<<<blocks, FFT_BLOCK, 0, c->stream>>>(real_in, spec, c->g.Ghp, c->twiddle);
    else k_row_c2r512<<<blocks, FFT_BLOCK, 0, c->stream>>>(spec, real_out, c->g.Ghp, c->twiddle);
    return hipGetLastError();
}

hipError_t pm_launch_col_fft_planes(pm_ctx *c, int dir, float2 *spec, int planes)
{
    const int G = c->g.G, P = c->g.Ghp;
    col_pass p{};
    p.spec = spec; p.G = G; p.ncols = G / 2 + 1; p.tiles = P / COLS;
    p.outer_stride = (int64_t)G * P; p.row_stride = P;
    if (dir < 0) return launch512<-1, false>(c, p, planes * p.tiles);
    return launch512<1, false>(c, p, planes * p.tiles);
}

// spec_t: (G, ny, Ghp), transform axis first; its rows are mesh rows y0 .. y0 + ny - 1
hipError_t pm_launch_solve_x(pm_ctx *c, float2 *spec_t, int y0, int ny)
{
    const int G = c->g.G, P = c->g.Ghp;
    col_pass p{};
    p.spec = spec_t; p.G = G; p.ncols = G / 2 + 1; p.tiles = P / COLS;
    p.invM = (float)(1.0 / (double)c->g.M);
    p.outer_stride = P; p.row_stride = (int64_t)ny * P; p.outer0 = y0;
    return launch512<-1, true>(c, p, ny * p.tiles);
}
