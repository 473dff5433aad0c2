// Counter-based random numbers for the kernels that draw particle sets: Philox4x32-10 (Salmon et al., "Parallel
// random numbers: as easy as 1, 2, 3", SC'11).  A draw is a pure function of (counter, key): a set does not depend on
// the launch shape.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pmrand {

struct u32x4 { uint32_t v[4]; };

__device__ __forceinline__ u32x4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{{c0, c1, c2, c3}};
}

// 32 random bits -> a double strictly inside (0, 1)
__device__ __forceinline__ double unit(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

// the stream of one particle: counter = (index, purpose, attempt), key = seed
struct draw_t {
    uint32_t lo, hi, k0, k1;
    __device__ u32x4 operator()(uint32_t purpose, uint32_t attempt) const { return philox(lo, hi, purpose, attempt, k0, k1); }
};

}  // namespace pmrand
