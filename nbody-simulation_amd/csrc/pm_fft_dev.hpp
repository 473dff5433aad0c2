// Device helpers of the hand-written FFTs, shared by the Poisson-solve kernels (pm_fft.hip) and the
// deposit kernels that transform a density row as soon as it is summed (pm_particles.hip).
#pragma once

#include <hip/hip_runtime.h>

namespace pmfft {

// The butterflies need no particular rounding.  Every function with a product allows it to fuse
// into the following add (PM_FFT_FUSE), whatever the including file is built with, so that a row
// transformed through either kernel has the same bits.
#define PM_FFT_FUSE _Pragma("clang fp contract(fast)")

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    PM_FFT_FUSE
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i (forward) or +i (inverse)
template <int DIR> __device__ __forceinline__ float2 rot90(float2 a)
{
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}
template <int DIR> __device__ __forceinline__ float2 twid(float2 w) { return DIR < 0 ? w : make_float2(w.x, -w.y); }

// 8-point DFT, in place, natural order in and out.  DIR = -1 forward (exp(-2 pi i nk/8)), +1 inverse.
template <int DIR> __device__ __forceinline__ void fft8(float2 *v)
{
    PM_FFT_FUSE
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

template <int DIR> __device__ __forceinline__ void fft4(float2 *v)
{
    float2 a0 = cadd(v[0], v[2]), a2 = csub(v[0], v[2]);
    float2 a1 = cadd(v[1], v[3]), a3 = rot90<DIR>(csub(v[1], v[3]));
    v[0] = cadd(a0, a1); v[2] = csub(a0, a1); v[1] = cadd(a2, a3); v[3] = csub(a2, a3);
}

// Orders the LDS accesses of one wave: what its lanes wrote before is visible to all its lanes
// after.  The LDS serves a wave's requests in issue order, so no instruction is needed, only that
// the compiler keeps the accesses on their side of this point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 256-point FFT of the values a wave holds as v[r] = element (lane + 64 r); result in the same form.
// `buf` is the wave's own 256-element LDS exchange buffer, `tw` the table exp(-2 pi i q / 512) in LDS,
// j the lane.  Radix-4 Stockham: one butterfly per lane per stage, every stage reads elements j + 64 r.
template <int DIR>
__device__ __forceinline__ void wave_fft256(float2 *v, float2 *buf, const float2 *tw, int j)
{
    fft4<DIR>(v);                                       // stage 1, Ns = 1: no twiddles
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[4 * j + r] = v[r];
#pragma unroll
    for (int s = 1; s < 4; ++s) {                       // Ns = 4, 16, 64
        const int Ns = 1 << (2 * s), k = j & (Ns - 1), step = 128 >> (2 * s);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = buf[j + 64 * r];
#pragma unroll
        for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], twid<DIR>(tw[r * k * step]));
        fft4<DIR>(v);
        if (s < 3) {
            wave_sync();
            const int j0 = ((j >> (2 * s)) << (2 * s + 2)) + k;
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[j0 + r * Ns] = v[r];
        }
    }
}

// A wave holds the 256-point transform Z of a packed real row (z[n] = x[2n] + i x[2n+1]) as
// v[r] = Z[j + 64 r]; writes the 257 values X[0..256] of the row's half spectrum to dst:
//   X[k] = (Z[k] + conj Z[256-k])/2 - (i/2) w^k (Z[k] - conj Z[256-k]),   w = exp(-2 pi i / 512).
__device__ __forceinline__ void wave_r2c_split512(const float2 *v, float2 *buf, const float2 *tw, int j, float2 *__restrict__ dst)
{
    PM_FFT_FUSE
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[j + 64 * r] = v[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = j + 64 * r;
        const float2 zk = v[r], zp = buf[(256 - k) & 255];
        const float2 e = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));      // (Z[k] + conj Z[256-k]) / 2
        const float2 d = make_float2(0.5f * (zk.x - zp.x), 0.5f * (zk.y + zp.y));      // (Z[k] - conj Z[256-k]) / 2
        const float2 t = cmul(tw[k], d);                                               // w^k d
        dst[k] = make_float2(e.x + t.y, e.y - t.x);                                    // e - i t
        if (k == 0) dst[256] = make_float2(zk.x - zk.y, 0.0f);
    }
}

}  // namespace pmfft
