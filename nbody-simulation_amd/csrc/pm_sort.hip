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

// ---- the same sort, driven from here ----------------------------------------------------------------------------
// rocprim::radix_sort_pairs clears its bookkeeping with seven small fill commands per sort (the digit counts, and per
// pass the look-back states and the ticket counter that numbers the blocks), each a launch of its own between the
// passes.  The driver below runs the library's own device code (histogram, scan, one-sweep pass: rocprim::detail) on a
// workspace that gives every pass its own look-back states and ticket counter, all cleared by ONE fill before the
// histogram kernel.  Same keys, same stable order.
#ifndef PM_OWN_ONESWEEP
#define PM_OWN_ONESWEEP 1
#endif

using ticket_t = rocprim::detail::block_id_wrapper<unsigned int, true>;     // blocks take their number from an atomic counter
using lookback_t = rocprim::detail::onesweep_lookback_state;

#ifndef PM_RUN_HIST
#define PM_RUN_HIST 1
#endif

#if PM_RUN_HIST
// Digit counts of every place in one pass over the keys.  The keys arrive in the cell order of the step before, so
// neighbouring keys mostly agree in their upper digits and often in all of them: a thread takes 16 consecutive keys at a time
// and adds one count per run of equal digits to the block's LDS table instead of one per key (the library's kernel,
// one LDS add per key and place, spends most of its 43 us at 16.7 M keys on adds that collide on one counter).
#ifndef PM_COUNT_BLOCK
#define PM_COUNT_BLOCK 1024
#endif
#ifndef PM_COUNT_GRID
#define PM_COUNT_GRID 512
#endif
constexpr unsigned COUNT_ITEMS = 16;
template <unsigned RADIX>
__global__ void __launch_bounds__(PM_COUNT_BLOCK)
k_digit_counts(const uint32_t *__restrict__ keys, uint32_t *__restrict__ counts, uint32_t size, uint32_t /*full_blocks*/, unsigned bits)
{
    constexpr unsigned ITEMS = COUNT_ITEMS, MAX_PLACES = (32 + RADIX - 1) / RADIX, SIZE = 1u << RADIX;
    static_assert(ITEMS % 4 == 0, "keys are loaded four at a time");
    __shared__ uint32_t table[MAX_PLACES * SIZE];
    const unsigned places = (bits + RADIX - 1) / RADIX;
    for (unsigned q = threadIdx.x; q < places * SIZE; q += PM_COUNT_BLOCK) table[q] = 0u;
    __syncthreads();
    // the blocks stride over the keys in chunks of PM_COUNT_BLOCK * ITEMS, so that a block's table is flushed once
    const uint64_t stride = (uint64_t)gridDim.x * PM_COUNT_BLOCK * ITEMS;
    for (uint64_t first = ((uint64_t)blockIdx.x * PM_COUNT_BLOCK + threadIdx.x) * ITEMS; first < size; first += stride) {
        uint32_t k[ITEMS];
        const unsigned have = size - first < ITEMS ? (unsigned)(size - first) : ITEMS;
        if (have == ITEMS) {
            const uint4 *src = (const uint4 *)(keys + first);
#pragma unroll
            for (unsigned q = 0; q < ITEMS / 4; ++q) {
                const uint4 v = src[q];
                k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (unsigned q = 0; q < ITEMS; ++q) k[q] = q < have ? keys[first + q] : 0u;
        }
        for (unsigned place = 0, bit = 0; place < places; ++place, bit += RADIX) {
            const uint32_t mask = (bits - bit < RADIX ? (1u << (bits - bit)) : SIZE) - 1u;
            uint32_t *row = table + place * SIZE;
            uint32_t cur = (k[0] >> bit) & mask, run = 1;
#pragma unroll
            for (unsigned q = 1; q < ITEMS; ++q) {
                const uint32_t d = (k[q] >> bit) & mask;
                if (q < have) {
                    if (d == cur) ++run;
                    else { atomicAdd(row + cur, run); cur = d; run = 1; }
                }
            }
            atomicAdd(row + cur, run);
        }
    }
    __syncthreads();
    for (unsigned q = threadIdx.x; q < places * SIZE; q += PM_COUNT_BLOCK) {
        const uint32_t v = table[q];
        if (v) atomicAdd(counts + q, v);
    }
}
#else
template <unsigned RADIX>
__global__ void __launch_bounds__(PM_HIST_BLOCK)
k_digit_counts(const uint32_t *keys, uint32_t *counts, uint32_t size, uint32_t full_blocks, unsigned bits)
{
    rocprim::detail::onesweep_histograms<PM_HIST_BLOCK, PM_HIST_ITEMS, RADIX, false>(keys, counts, size, full_blocks, rocprim::identity_decomposer{}, 0u, bits);
}
#endif

template <unsigned RADIX>
__global__ void __launch_bounds__(PM_HIST_BLOCK)
k_digit_offsets(uint32_t *counts)
{
    rocprim::detail::onesweep_scan_histograms<PM_HIST_BLOCK, RADIX>(counts);
}

// VIn: where a pass reads its values -- the scratch or output array, or for the first pass a counting iterator (the
// values that go in are the storage indices 0, 1, 2, ...: nothing to read)
template <unsigned RADIX, class VIn>
__global__ void __launch_bounds__(PM_SORT_BLOCK)
k_onesweep_pass(const uint32_t *kin, uint32_t *kout, VIn vin, uint32_t *vout, uint32_t size, uint32_t *offsets_in, uint32_t *offsets_out,
                lookback_t *lookback, unsigned bit, unsigned place_bits, uint32_t full_blocks, ticket_t ticket)
{
    rocprim::detail::onesweep_iteration<PM_SORT_BLOCK, PM_SORT_ITEMS, RADIX, false, rocprim::block_radix_rank_algorithm::match>(
        kin, kout, vin, vout, size, offsets_in, offsets_out, lookback, rocprim::identity_decomposer{}, bit, place_bits, full_blocks, ticket);
}

constexpr size_t SORT_ALIGN = 256;
inline size_t aligned(size_t b) { return (b + SORT_ALIGN - 1) / SORT_ALIGN * SORT_ALIGN; }

// workspace: [tickets | look-back states of every pass | digit counts of every pass | spare offsets] (cleared per sort),
// then the scratch keys and values
struct sort_plan {
    unsigned radix, places, blocks;
    size_t tickets, lookback, counts, spare, cleared, keys_tmp, values_tmp, bytes;
};

template <unsigned RADIX>
sort_plan plan_sort(size_t n, unsigned bits)
{
    sort_plan p{};
    p.radix = 1u << RADIX;
    p.places = (bits + RADIX - 1) / RADIX;
    p.blocks = (unsigned)((n + (size_t)PM_SORT_BLOCK * PM_SORT_ITEMS - 1) / ((size_t)PM_SORT_BLOCK * PM_SORT_ITEMS));
    p.tickets = 0;
    p.lookback = aligned(p.places * sizeof(uint32_t));
    p.counts = p.lookback + aligned((size_t)p.places * p.radix * p.blocks * sizeof(lookback_t));
    p.spare = p.counts + aligned((size_t)p.places * p.radix * sizeof(uint32_t));
    p.cleared = p.spare + aligned((size_t)p.radix * sizeof(uint32_t));
    p.keys_tmp = p.cleared;
    p.values_tmp = p.keys_tmp + aligned(n * sizeof(uint32_t));
    p.bytes = p.values_tmp + aligned(n * sizeof(uint32_t));
    return p;
}

template <unsigned RADIX>
hipError_t own_sort_pairs(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned bits,
                          hipStream_t s)
{
    const sort_plan p = plan_sort<RADIX>(n, bits);
    if (p.bytes > tmp_bytes || n >= ((size_t)1 << 30)) return hipErrorOutOfMemory;       // one batch of the library's pass: below 2^30 pairs
    char *w = (char *)tmp;
    uint32_t *tickets = (uint32_t *)(w + p.tickets), *counts = (uint32_t *)(w + p.counts), *spare = (uint32_t *)(w + p.spare);
    lookback_t *lookback = (lookback_t *)(w + p.lookback);
    uint32_t *ktmp = (uint32_t *)(w + p.keys_tmp), *vtmp = (uint32_t *)(w + p.values_tmp);
    hipError_t e = hipMemsetAsync(w, 0, p.cleared, s);
    if (e != hipSuccess) return e;
#if PM_RUN_HIST
    const size_t per_count = (size_t)PM_COUNT_BLOCK * COUNT_ITEMS;
    const size_t want = (n + per_count - 1) / per_count;
    k_digit_counts<RADIX><<<(unsigned)(want < PM_COUNT_GRID ? want : PM_COUNT_GRID), PM_COUNT_BLOCK, 0, s>>>(kin, counts, (uint32_t)n, 0u, bits);
#else
    const size_t per_hist = (size_t)PM_HIST_BLOCK * PM_HIST_ITEMS;
    const uint32_t hist_blocks = (uint32_t)((n + per_hist - 1) / per_hist), hist_full = (uint32_t)(n / per_hist);
    k_digit_counts<RADIX><<<hist_blocks, PM_HIST_BLOCK, 0, s>>>(kin, counts, (uint32_t)n, hist_full, bits);
#endif
    k_digit_offsets<RADIX><<<p.places, PM_HIST_BLOCK, 0, s>>>(counts);
    const size_t per_block = (size_t)PM_SORT_BLOCK * PM_SORT_ITEMS;
    const uint32_t full_blocks = (uint32_t)(n / per_block);
    // the last pass must end in the output: with an odd number of passes the first one goes there too
    bool to_output = (p.places - 1) % 2 == 0, from_input = true;
    for (unsigned place = 0, bit = 0; place < p.places; ++place, bit += RADIX) {
        const uint32_t *ki = from_input ? kin : to_output ? ktmp : kout, *vi = from_input ? vin : to_output ? vtmp : vout;
        uint32_t *ko = to_output ? kout : ktmp, *vo = to_output ? vout : vtmp;
        const unsigned now = bits - bit < RADIX ? bits - bit : RADIX;
        uint32_t *offsets = counts + (size_t)place * p.radix;
        lookback_t *states = lookback + (size_t)place * p.radix * p.blocks;
        const ticket_t ticket = ticket_t::create(tickets + place);
#ifndef PM_SORT_READ_IOTA
        if (from_input && vin == nullptr)
            k_onesweep_pass<RADIX><<<p.blocks, PM_SORT_BLOCK, 0, s>>>(ki, ko, rocprim::counting_iterator<uint32_t>(0), vo, (uint32_t)n, offsets, spare, states,
                                                                      bit, now, full_blocks, ticket);
        else
#endif
            k_onesweep_pass<RADIX><<<p.blocks, PM_SORT_BLOCK, 0, s>>>(ki, ko, vi, vo, (uint32_t)n, offsets, spare, states, bit, now, full_blocks, ticket);
        from_input = false;
        to_output = !to_output;
    }
    return hipGetLastError();
}

__global__ void k_iota(uint32_t *__restrict__ out, int64_t n)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) out[r] = (uint32_t)r;
}

// The fewest places wins: 10 bits per place where that saves a place over 9 and 8 (the 30 key bits of a 1024^3 mesh in
// three passes instead of four: 2.43 -> 2.07 ms for the bin stage at 67 M particles), else 9 where that saves one over
// the library's 8 (27 bits of a 512^3 mesh: three passes), else 8.
unsigned place_bits(unsigned bits)
{
#ifndef PM_NO_RADIX10
    if ((bits + 9) / 10 < (bits + 8) / 9 && (bits + 9) / 10 < (bits + 7) / 8) return 10;
#endif
    return (bits + 8) / 9 < (bits + 7) / 8 ? 9 : 8;
}

template <typename F> hipError_t with_radix(unsigned bits, F f)
{
    switch (place_bits(bits)) {
    case 10: return f(std::integral_constant<unsigned, 10>{});
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
        // the own driver: 32 key bits at most, in places of 8 or 9 bits
        for (unsigned kb = 1; kb <= 32; ++kb) {
            with_radix(kb, [&](auto r) { b = plan_sort<decltype(r)::value>((size_t)cap, kb).bytes; return hipSuccess; });
            most = b > most ? b : most;
        }
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
#if PM_OWN_ONESWEEP
    // (2^30 pairs and more go through the library's call below, which cuts its passes into batches)
    if (c->n < ((int64_t)1 << 30)) return with_radix((unsigned)bits, [&](auto r) {
        // values in: none (the storage indices are counted, not read); PM_SORT_READ_IOTA reads them from the table instead
#ifdef PM_SORT_READ_IOTA
        const uint32_t *vin = c->iota;
#else
        const uint32_t *vin = nullptr;
#endif
        return own_sort_pairs<decltype(r)::value>(c->sort_tmp, c->sort_tmp_bytes, c->key_in, c->key_out, vin, c->src, (size_t)c->n, (unsigned)bits, c->stream);
    });
#endif
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
