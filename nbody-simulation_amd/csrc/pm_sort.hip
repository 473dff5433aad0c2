// Stable LSD radix sort of (cell key, storage index) pairs.  Stability is what makes the
// cell order well defined: ties keep the current storage order.
#include "pm_internal.hpp"

#include <cstring>
#include <rocprim/rocprim.hpp>

size_t pm_sort_tmp_bytes(int64_t cap)
{
    size_t bytes = 0;
    uint32_t *k = nullptr, *v = nullptr;
    rocprim::radix_sort_pairs(nullptr, bytes, k, k, rocprim::counting_iterator<uint32_t>(0), v, (size_t)cap, 0, 32, (hipStream_t)0);
    return bytes;
}

hipError_t pm_sort_pairs(pm_ctx *c, int bits)
{
    if (c->n == 0) return hipSuccess;
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, c->key_in, c->key_out, rocprim::counting_iterator<uint32_t>(0), c->src,
                                             (size_t)c->n, 0, (unsigned)bits, c->stream);
    if (e != hipSuccess) return e;
    if (need > c->sort_tmp_bytes) return hipErrorOutOfMemory;
    return rocprim::radix_sort_pairs(c->sort_tmp, need, c->key_in, c->key_out, rocprim::counting_iterator<uint32_t>(0), c->src,
                                     (size_t)c->n, 0, (unsigned)bits, c->stream);
}
