// Stable LSD radix sort of (cell key, storage index) pairs.  Stability is what makes the
// cell order well defined: ties keep the current storage order.
#include "pm_internal.hpp"

#include <cstring>
#include <rocprim/rocprim.hpp>

// block size and keys per thread of the histogram kernel that precedes the passes (16 keys per thread: 14 us less
// than with 8 at 16.7 M keys; more is slower again)
#ifndef PM_HIST_BLOCK
#define PM_HIST_BLOCK 1024
#endif
#ifndef PM_HIST_ITEMS
#define PM_HIST_ITEMS 16
#endif
#ifndef PM_SORT_BLOCK
#define PM_SORT_BLOCK 1024
#endif
#ifndef PM_SORT_ITEMS
#define PM_SORT_ITEMS 8
#endif

namespace {

// One pass over the pairs per "place" of RADIX bits.  The library's tuned default is 8 bits; with
// 9 the 27 key bits of a 512^3 mesh take three passes instead of four.  1024 threads x 8 items of
// 4 + 4 bytes fill the 64 KiB of LDS a block may use.
template <unsigned RADIX>
using sort_config = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<PM_HIST_BLOCK, PM_HIST_ITEMS>, rocprim::kernel_config<PM_SORT_BLOCK, PM_SORT_ITEMS>, RADIX,
                                        rocprim::block_radix_rank_algorithm::match>>;

// The values that go in are the storage indices 0, 1, 2, ...: read from a table that is written once, not made by a
// counting iterator, because the library cannot tell that an iterator does not alias the output and then copies keys and
// values into its scratch before the first pass (two more kernels, 32 us at 16.7 M pairs).
template <unsigned RADIX>
hipError_t sort_pairs(void *tmp, size_t &bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned bits,
                      hipStream_t s)
{
    return rocprim::radix_sort_pairs<sort_config<RADIX>>(tmp, bytes, kin, kout, vin, vout, n, 0, bits, s);
}

__global__ void k_iota(uint32_t *__restrict__ out, int64_t n)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) out[r] = (uint32_t)r;
}

// 9 bits per place where that saves a place, else the library's 8 (10-bit places measured slower than one more 8-bit place)
unsigned place_bits(unsigned bits) { return (bits + 8) / 9 < (bits + 7) / 8 ? 9 : 8; }

template <typename F> hipError_t with_radix(unsigned bits, F f)
{
    switch (place_bits(bits)) {
    case 9: return f(std::integral_constant<unsigned, 9>{});
    default: return f(std::integral_constant<unsigned, 8>{});
    }
}

}  // namespace

size_t pm_sort_tmp_bytes(int64_t cap)
{
    size_t most = 0;
    uint32_t *k = nullptr, *v = nullptr;
    for (unsigned bits : {8u, 27u}) {
        size_t b = 0;
        with_radix(bits, [&](auto r) { return sort_pairs<decltype(r)::value>(nullptr, b, k, k, v, v, (size_t)cap, 32, (hipStream_t)0); });
        most = b > most ? b : most;
    }
    return most;
}

hipError_t pm_sort_fill_iota(pm_ctx *c)
{
    k_iota<<<(unsigned)((c->cap + 255) / 256), 256, 0, c->stream>>>(c->iota, c->cap);
    return hipGetLastError();
}

hipError_t pm_sort_pairs(pm_ctx *c, int bits)
{
    if (c->n == 0) return hipSuccess;
    size_t need = 0;
    auto run = [&](void *tmp) {
        return with_radix((unsigned)bits, [&](auto r) {
            return sort_pairs<decltype(r)::value>(tmp, need, c->key_in, c->key_out, c->iota, c->src, (size_t)c->n, (unsigned)bits, c->stream);
        });
    };
    hipError_t e = run(nullptr);
    if (e != hipSuccess) return e;
    if (need > c->sort_tmp_bytes) return hipErrorOutOfMemory;
    return run(c->sort_tmp);
}

// Particles leaving a slab: (destination slab, particle id) keys with the outbox slot as value.
size_t pm_sort_outbox_tmp_bytes(int64_t cap)
{
    size_t bytes = 0;
    uint64_t *k = nullptr;
    uint32_t *v = nullptr;
    rocprim::radix_sort_pairs(nullptr, bytes, k, k, rocprim::counting_iterator<uint32_t>(0), v, (size_t)cap, 0, 64, (hipStream_t)0);
    return bytes;
}

hipError_t pm_sort_outbox(pm_ctx *c, uint32_t count)
{
    if (count == 0) return hipSuccess;
    size_t need = c->outbox_tmp_bytes;
    return rocprim::radix_sort_pairs(c->outbox_tmp, need, c->outbox_key, c->outbox_key2, rocprim::counting_iterator<uint32_t>(0), c->outbox_idx2,
                                     (size_t)count, 0, 64, c->stream);
}
