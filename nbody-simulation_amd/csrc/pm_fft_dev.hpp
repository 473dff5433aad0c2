// Device helpers of the hand-written FFTs, shared by the Poisson-solve kernels (pm_fft.hip) and the
// deposit kernels that transform a density row as soon as it is summed (pm_particles.hip).
#pragma once

#include <hip/hip_runtime.h>

namespace pmfft {

// The butterflies need no particular rounding.  Every function with a product allows it to fuse
// into the following add (PM_FFT_FUSE), whatever the including file is built with, so that a row
// transformed through either kernel has the same bits.
#define PM_FFT_FUSE _Pragma("clang fp contract(fast)")

// (the adds too: a product made in one helper fuses into the add made in another only if both allow it)
__device__ __forceinline__ float2 cadd(float2 a, float2 b)
{
    PM_FFT_FUSE
    return make_float2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ float2 csub(float2 a, float2 b)
{
    PM_FFT_FUSE
    return make_float2(a.x - b.x, a.y - b.y);
}
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

// 16-point DFT, in place, natural order in and out: 4 x 4 with the twiddles w16^(r c) between the two passes.
template <int DIR> __device__ __forceinline__ void fft16(float2 *v)
{
    PM_FFT_FUSE
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    float2 t[4][4];                                     // t[c][r]: column c = n mod 4 holds n = c + 4 r
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[c][r] = v[c + 4 * r];
        fft4<DIR>(t[c]);                                // t[c][k1], k1 = 0..3
    }
    // t[c][k1] *= w16^(c k1), w16 = exp(DIR * 2 pi i / 16)
    const float2 w1 = make_float2(c1, DIR < 0 ? -s1 : s1), w2 = make_float2(h, DIR < 0 ? -h : h), w3 = make_float2(s1, DIR < 0 ? -c1 : c1);
    const float2 w6 = make_float2(-h, DIR < 0 ? -h : h), w9 = make_float2(-c1, DIR < 0 ? s1 : -s1);
    t[1][1] = cmul(t[1][1], w1); t[1][2] = cmul(t[1][2], w2); t[1][3] = cmul(t[1][3], w3);
    t[2][1] = cmul(t[2][1], w2); t[2][2] = rot90<DIR>(t[2][2]); t[2][3] = cmul(t[2][3], w6);
    t[3][1] = cmul(t[3][1], w3); t[3][2] = cmul(t[3][2], w6); t[3][3] = cmul(t[3][3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float2 u[4] = {t[0][k1], t[1][k1], t[2][k1], t[3][k1]};
        fft4<DIR>(u);                                   // u[k2]: output k = k1 + 4 k2
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = u[k2];
    }
}

// 32-point DFT, in place, natural order in and out: 4 columns of 8 (n = c + 4 r), the twiddles w32^(c k1), then 8
// transforms of 4 (output k = k1 + 8 k2).
template <int DIR> __device__ __forceinline__ void fft32(float2 *v)
{
    PM_FFT_FUSE
    // cos and sin of 2 pi m / 32, m = 0 .. 21 (the largest c k1 is 3 * 7)
    constexpr float C32[22] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                               0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f, -0.19509032201612826785f,
                               -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                               -0.92387953251128675613f, -0.98078528040323044913f, -1.0f, -0.98078528040323044913f, -0.92387953251128675613f,
                               -0.83146961230254523708f, -0.70710678118654752440f, -0.55557023301960222474f};
    constexpr float S32[22] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                               0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.0f, 0.98078528040323044913f,
                               0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                               0.38268343236508977173f, 0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                               -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f};
    float2 t[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r = 0; r < 8; ++r) t[c][r] = v[c + 4 * r];
        fft8<DIR>(t[c]);
    }
#pragma unroll
    for (int c = 1; c < 4; ++c)
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) {
            const int m = c * k1;                       // w32^m = exp(DIR * 2 pi i m / 32)
            t[c][k1] = cmul(t[c][k1], make_float2(C32[m], DIR < 0 ? -S32[m] : S32[m]));
        }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
        float2 u[4] = {t[0][k1], t[1][k1], t[2][k1], t[3][k1]};
        fft4<DIR>(u);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) v[k1 + 8 * k2] = u[k2];
    }
}

template <int R, int DIR> __device__ __forceinline__ void fft_r(float2 *v)
{
    if constexpr (R == 2) { const float2 a = v[0], b = v[1]; v[0] = cadd(a, b); v[1] = csub(a, b); }
    else if constexpr (R == 4) fft4<DIR>(v);
    else if constexpr (R == 8) fft8<DIR>(v);
    else if constexpr (R == 16) fft16<DIR>(v);
    else fft32<DIR>(v);
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

// 64-point FFT of one value per lane: radix 2, six Stockham stages through the wave's LDS buffer.  A stage has 32
// butterflies; both halves of the wave work each of them out and each half writes one of its two results.
template <int DIR>
__device__ __forceinline__ void wave_fft64(float2 &v, float2 *buf, const float2 *tw, int j)
{
    const int b = j & 31;
    const bool upper = j >= 32;
    wave_sync();
    buf[j] = v;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int Ns = 1 << s, k = b & (Ns - 1);
        wave_sync();
        const float2 x0 = buf[b];
        const float2 x1 = cmul(buf[b + 32], twid<DIR>(tw[k * (64 >> s)]));          // exp(-+ 2 pi i k / (2 Ns)) from the table of 128
        const float2 y = upper ? csub(x0, x1) : cadd(x0, x1);
        wave_sync();
        buf[((b - k) << 1) + k + (upper ? Ns : 0)] = y;
    }
    wave_sync();
    v = buf[j];
}

// M-point FFT, M = 64 E, of the values a wave holds as v[r] = element (lane + 64 r), r < E; result in
// the same form.  `buf` is the wave's own M-element LDS exchange buffer, `tw` the table
// exp(-2 pi i q / (2 M)) in LDS, j the lane.  Radix-E Stockham (E = 2: seven stages, E = 4: four, E = 8: three):
// one butterfly per lane per stage, every stage reads elements j + 64 r.  E = 1 is the 64-point transform above.
// 1024-point FFT of the values a wave holds as v[r] = element (lane + 64 r), r < 16: radix 16, 16, 4.  The first two
// stages are one butterfly per lane as below; the last has 256 butterflies of 4, four per lane: butterfly b = lane + 64 q
// reads elements b + 256 r and leaves element b + 256 r of the transform, i.e. v[q + 4 r].
template <int DIR>
__device__ __forceinline__ void wave_fft1024(float2 *v, float2 *buf, const float2 *tw, int j)
{
    fft16<DIR>(v);                                      // stage 1, Ns = 1
    wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[16 * j + r] = v[r];
    {
        const int k = j & 15;                           // stage 2, Ns = 16: twiddles exp(-+ 2 pi i r k / 256) from the table of 2048
        wave_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = buf[j + 64 * r];
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], twid<DIR>(tw[8 * r * k]));
        fft16<DIR>(v);
        wave_sync();
        const int j0 = ((j >> 4) << 8) + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[j0 + 16 * r] = v[r];
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) {                       // stage 3, Ns = 256: twiddles exp(-+ 2 pi i r b / 1024)
        const int b = j + 64 * q;
        float2 u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = buf[b + 256 * r];
#pragma unroll
        for (int r = 1; r < 4; ++r) u[r] = cmul(u[r], twid<DIR>(tw[2 * r * b]));
        fft4<DIR>(u);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[q + 4 * r] = u[r];
    }
}

template <int E, int DIR>
__device__ __forceinline__ void wave_fft(float2 *v, float2 *buf, const float2 *tw, int j)
{
    if constexpr (E == 1) { wave_fft64<DIR>(v[0], buf, tw, j); return; }
    if constexpr (E == 16) { wave_fft1024<DIR>(v, buf, tw, j); return; }
    constexpr int LOG = E == 2 ? 1 : E == 4 ? 2 : 3, STAGES = E == 2 ? 7 : E == 4 ? 4 : 3;
    fft_r<E, DIR>(v);                                   // stage 1, Ns = 1: no twiddles
    wave_sync();
#pragma unroll
    for (int r = 0; r < E; ++r) buf[E * j + r] = v[r];
#pragma unroll
    for (int s = 1; s < STAGES; ++s) {                  // Ns = E, E^2, ...
        const int Ns = 1 << (LOG * s), k = j & (Ns - 1), step = (2 * 64) >> (LOG * s);     // 2 M / (Ns E)
        wave_sync();
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = buf[j + 64 * r];
#pragma unroll
        for (int r = 1; r < E; ++r) v[r] = cmul(v[r], twid<DIR>(tw[r * k * step]));
        fft_r<E, DIR>(v);
        if (s < STAGES - 1) {
            wave_sync();
            const int j0 = ((j >> (LOG * s)) << (LOG * s + LOG)) + k;
#pragma unroll
            for (int r = 0; r < E; ++r) buf[j0 + r * Ns] = v[r];
        }
    }
}

template <int DIR>
__device__ __forceinline__ void wave_fft256(float2 *v, float2 *buf, const float2 *tw, int j) { wave_fft<4, DIR>(v, buf, tw, j); }

// A wave holds the M-point transform Z of a packed real row of 2 M values (z[n] = x[2n] + i x[2n+1]) as
// v[r] = Z[j + 64 r]; writes the M + 1 values X[0..M] of the row's half spectrum to dst:
//   X[k] = (Z[k] + conj Z[M-k])/2 - (i/2) w^k (Z[k] - conj Z[M-k]),   w = exp(-2 pi i / (2 M)).
template <int E>
__device__ __forceinline__ void wave_r2c_split(const float2 *v, float2 *buf, const float2 *tw, int j, float2 *__restrict__ dst)
{
    PM_FFT_FUSE
    constexpr int M = 64 * E;
    wave_sync();
#pragma unroll
    for (int r = 0; r < E; ++r) buf[j + 64 * r] = v[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int k = j + 64 * r;
        const float2 zk = v[r], zp = buf[(M - k) & (M - 1)];
        const float2 e = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));      // (Z[k] + conj Z[M-k]) / 2
        const float2 d = make_float2(0.5f * (zk.x - zp.x), 0.5f * (zk.y + zp.y));      // (Z[k] - conj Z[M-k]) / 2
        const float2 t = cmul(tw[k], d);                                               // w^k d
        dst[k] = make_float2(e.x + t.y, e.y - t.x);                                    // e - i t
        if (k == 0) dst[M] = make_float2(zk.x - zk.y, 0.0f);
    }
}

__device__ __forceinline__ void wave_r2c_split512(const float2 *v, float2 *buf, const float2 *tw, int j, float2 *__restrict__ dst)
{
    wave_r2c_split<4>(v, buf, tw, j, dst);
}

// The inverse of wave_r2c_split: the lanes hold x[r] = X[j + 64 r] and xm = X[M] (lane 0) of a half
// spectrum; returns Z with z = the unnormalised inverse M-point FFT still to be taken:
//   Z[k] = (X[k] + conj X[M-k]) + i w^-k (X[k] - conj X[M-k]).   `buf` holds M + 1 elements.
template <int E>
__device__ __forceinline__ void wave_c2r_merge(const float2 *x, float2 xm, float2 *v, float2 *buf, const float2 *tw, int j)
{
    PM_FFT_FUSE
    constexpr int M = 64 * E;
    wave_sync();
#pragma unroll
    for (int r = 0; r < E; ++r) buf[j + 64 * r] = x[r];
    if (j == 0) buf[M] = xm;
    wave_sync();
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int k = j + 64 * r;
        float2 xk = x[r], xp = buf[M - k];
        if (k == 0) { xk.y = 0.0f; xp.y = 0.0f; }                      // X[0] and X[M] of a real row are real
        const float2 e = make_float2(xk.x + xp.x, xk.y - xp.y);        // X[k] + conj X[M-k]
        const float2 d = make_float2(xk.x - xp.x, xk.y + xp.y);        // X[k] - conj X[M-k]
        const float2 wk = tw[k];
        const float2 t = cmul(make_float2(wk.x, -wk.y), d);            // w^-k d
        v[r] = make_float2(e.x - t.y, e.y + t.x);                      // e + i t
    }
}

}  // namespace pmfft
