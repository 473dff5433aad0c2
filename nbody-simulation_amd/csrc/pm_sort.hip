// Stable LSD radix sort of (cell key, storage index) pairs for gfx950, hand-written.  Stability is what makes the cell
// order well defined: ties keep the current storage order (functions.py:27-39 defines the cells, the order of equal
// cells is the order the particles are stored in).
//
// A place (digit) of the keys takes one pass that ranks the pairs of a tile of 8192 by their digit and writes them to
// their places.  Where a tile's share of each digit goes follows from the digit counts of the tiles before it, and there
// are two ways to learn them:
//  - a sort whose number of pairs the host knows: a counting kernel writes every tile's digit counts, a scan turns them
//    into each tile's first place per digit, and the pass reads its row of that table.  No tile waits for another;
//  - a sort whose number of pairs is known on the device only (the movers of pm_rebin.hip): one kernel counts the digits
//    of every place, one turns the counts into offsets, and a tile learns its places from the tiles before it through a
//    chain of look-back states inside the pass ("one sweep").  Fewer launches, which is what a sort of few pairs is
//    made of; on 16.7 M pairs a pass spends 36 of its 92 us in that chain, which is why the other way exists.
//
// (Measured and not kept: a last pass that also fetches the position record of every pair and writes it, with its
// fraction record, at the pair's place in cell order.  It saves reading the sorted indices again but holds a tile's
// LDS and 86 registers per thread while the scattered reads are in flight: 303 us against 107 + 178 us for the pass
// and the separate reorder kernel at 16.7 M particles.)
#include "pm_internal.hpp"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {

#ifndef PM_SORT_BLOCK
#define PM_SORT_BLOCK 1024
#endif
constexpr int SORT_BLOCK = PM_SORT_BLOCK;           // threads of a pass's workgroup: 16 waves
constexpr int SORT_WAVES = SORT_BLOCK / 64;
#ifndef PM_SORT_ITEMS
#define PM_SORT_ITEMS 8
#endif
constexpr int SORT_ITEMS = PM_SORT_ITEMS;           // pairs per thread
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;  // pairs per workgroup

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

template <typename K> __device__ __forceinline__ uint32_t digit_of(K key, unsigned bit, uint32_t mask) { return (uint32_t)(key >> bit) & mask; }

// ---- digit counts of every place in one pass over the keys --------------------------------------------------------
// The keys of a step arrive in the cell order of the step before, so neighbouring keys mostly agree in their upper
// digits and often in all of them: a thread takes 16 consecutive keys at a time and adds one count per run of equal
// digits to the block's LDS table instead of one per key (one LDS add per key and place spends most of its time on
// adds that collide on one counter).
#ifndef PM_COUNT_BLOCK
#define PM_COUNT_BLOCK 1024
#endif
#ifndef PM_COUNT_GRID
#define PM_COUNT_GRID 512
#endif
constexpr unsigned COUNT_ITEMS = 16;

template <typename K, unsigned RADIX>
__global__ void __launch_bounds__(PM_COUNT_BLOCK)
k_digit_counts(const K *__restrict__ keys, uint32_t *__restrict__ counts, uint32_t size, unsigned bits, const uint32_t *__restrict__ size_dev)
{
    if (size_dev) size = *size_dev;                     // the number of keys is known on the device only (at most the `size` given)
    constexpr unsigned ITEMS = COUNT_ITEMS, MAX_PLACES = (8 * sizeof(K) + RADIX - 1) / RADIX, SIZE = 1u << RADIX;
    static_assert(ITEMS % 4 == 0, "keys are loaded four at a time");
    extern __shared__ uint32_t table[];                 // places * SIZE counters
    const unsigned places = (bits + RADIX - 1) / RADIX;
    if (places > MAX_PLACES) return;
    for (unsigned q = threadIdx.x; q < places * SIZE; q += PM_COUNT_BLOCK) table[q] = 0u;
    __syncthreads();
    // the blocks stride over the keys in chunks of PM_COUNT_BLOCK * ITEMS, so that a block's table is flushed once
    const uint64_t stride = (uint64_t)gridDim.x * PM_COUNT_BLOCK * ITEMS;
    for (uint64_t first = ((uint64_t)blockIdx.x * PM_COUNT_BLOCK + threadIdx.x) * ITEMS; first < size; first += stride) {
        K k[ITEMS];
        const unsigned have = size - first < ITEMS ? (unsigned)(size - first) : ITEMS;
        if (have == ITEMS) {
            if constexpr (sizeof(K) == 4) {
                const uint4 *src = (const uint4 *)(keys + first);
#pragma unroll
                for (unsigned q = 0; q < ITEMS / 4; ++q) {
                    const uint4 v = src[q];
                    k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (unsigned q = 0; q < ITEMS; ++q) k[q] = keys[first + q];
            }
        } else {
#pragma unroll
            for (unsigned q = 0; q < ITEMS; ++q) k[q] = q < have ? keys[first + q] : (K)0;
        }
        for (unsigned place = 0, bit = 0; place < places; ++place, bit += RADIX) {
            const uint32_t mask = (bits - bit < RADIX ? (1u << (bits - bit)) : SIZE) - 1u;
            uint32_t *row = table + place * SIZE;
            uint32_t cur = digit_of(k[0], bit, mask), run = 1;
#pragma unroll
            for (unsigned q = 1; q < ITEMS; ++q) {
                const uint32_t d = digit_of(k[q], bit, mask);
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

// Exclusive sum over the threads of a workgroup of SORT_BLOCK threads (every thread must call it); wsum: SORT_WAVES words of LDS.
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t *wsum)
{
    const int lane = lane_id(), w = wave_id();
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int q = 0; q < w; ++q) before += wsum[q];
    __syncthreads();                                    // wsum may be reused by the caller's next scan
    return before + incl - v;
}

// counts[place][digit] -> number of keys whose digit at that place is smaller: one workgroup per place
template <unsigned RADIX>
__global__ void __launch_bounds__(SORT_BLOCK)
k_digit_offsets(uint32_t *__restrict__ counts)
{
    constexpr uint32_t SIZE = 1u << RADIX;
    static_assert(SIZE <= SORT_BLOCK, "a thread per digit");
    __shared__ uint32_t wsum[SORT_WAVES];
    uint32_t *row = counts + (size_t)blockIdx.x * SIZE;
    const uint32_t v = threadIdx.x < SIZE ? row[threadIdx.x] : 0u;
    const uint32_t before = block_exclusive_sum(v, wsum);
    if (threadIdx.x < SIZE) row[threadIdx.x] = before;
}

// ---- digit counts of every tile for ONE place, and their scan ------------------------------------------------------
// tile_counts[tile][digit]: pairs of the tile (SORT_TILE consecutive pairs of the pass's input) with that digit.  A
// thread takes SORT_ITEMS consecutive keys and adds one count per run of equal digits.
template <typename K, unsigned RADIX>
__global__ void __launch_bounds__(SORT_BLOCK)
k_tile_counts(const K *__restrict__ keys, uint32_t n, uint32_t *__restrict__ tile_counts, unsigned bit, uint32_t mask)
{
    constexpr uint32_t SIZE = 1u << RADIX;
    static_assert(SORT_ITEMS % 4 == 0, "keys are loaded four at a time");
    __shared__ uint32_t table[SIZE];
    for (uint32_t q = threadIdx.x; q < SIZE; q += SORT_BLOCK) table[q] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)SORT_TILE + threadIdx.x * (uint32_t)SORT_ITEMS;
    if (base < n) {
        const uint32_t have = n - base < (uint32_t)SORT_ITEMS ? n - base : (uint32_t)SORT_ITEMS;
        K k[SORT_ITEMS];
        if (have == (uint32_t)SORT_ITEMS) {
            if constexpr (sizeof(K) == 4) {
                const uint4 *src = (const uint4 *)(keys + base);
#pragma unroll
                for (int q = 0; q < SORT_ITEMS / 4; ++q) {
                    const uint4 v = src[q];
                    k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < SORT_ITEMS; ++q) k[q] = keys[base + q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < SORT_ITEMS; ++q) k[q] = (uint32_t)q < have ? keys[base + q] : (K)0;
        }
        uint32_t cur = digit_of(k[0], bit, mask), run = 1;
#pragma unroll
        for (int q = 1; q < SORT_ITEMS; ++q) {
            const uint32_t d = digit_of(k[q], bit, mask);
            if ((uint32_t)q < have) {
                if (d == cur) ++run;
                else { atomicAdd(table + cur, run); cur = d; run = 1; }
            }
        }
        atomicAdd(table + cur, run);
    }
    __syncthreads();
    uint32_t *out = tile_counts + (size_t)blockIdx.x * SIZE;
    for (uint32_t q = threadIdx.x; q < SIZE; q += SORT_BLOCK) out[q] = table[q];
}

// The scan of that table over the tiles, in place: tile_counts[tile][digit] -> pairs with that digit in the tiles
// before `tile`; digit_total[digit] = pairs with that digit in all tiles.  Two kernels, every access a whole row of the
// table: the tiles are cut into at most SCAN_CHUNKS chunks of consecutive tiles; k_chunk_sums adds up the rows of each
// chunk; k_scan_chunks gives a workgroup a chunk, has it add the sums of the chunks before its own (a thread per digit)
// and walk its tiles with the running count.
constexpr uint32_t SCAN_CHUNKS = 128;
template <unsigned RADIX>
__global__ void __launch_bounds__(1u << RADIX)
k_chunk_sums(const uint32_t *__restrict__ tile_counts, uint32_t tiles, uint32_t per_chunk, uint32_t *__restrict__ chunk_sums)
{
    constexpr uint32_t SIZE = 1u << RADIX;
    const uint32_t d = threadIdx.x, t0 = blockIdx.x * per_chunk, t1 = t0 + per_chunk < tiles ? t0 + per_chunk : tiles;
    uint32_t sum = 0;
    uint32_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = tile_counts[(size_t)(t + q) * SIZE + d];
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += v[q];
    }
    for (; t < t1; ++t) sum += tile_counts[(size_t)t * SIZE + d];
    chunk_sums[(size_t)blockIdx.x * SIZE + d] = sum;
}

template <unsigned RADIX>
__global__ void __launch_bounds__(1u << RADIX)
k_scan_chunks(uint32_t *__restrict__ tile_counts, uint32_t tiles, uint32_t per_chunk, const uint32_t *__restrict__ chunk_sums,
              uint32_t *__restrict__ digit_total)
{
    constexpr uint32_t SIZE = 1u << RADIX;
    const uint32_t d = threadIdx.x, t0 = blockIdx.x * per_chunk, t1 = t0 + per_chunk < tiles ? t0 + per_chunk : tiles;
    // the first workgroup also adds up all chunks: the digit's total
    const uint32_t upto = blockIdx.x == 0 ? gridDim.x : blockIdx.x;
    uint32_t before = 0;
    uint32_t c = 0;
    for (; c + 8 <= upto; c += 8) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = chunk_sums[(size_t)(c + q) * SIZE + d];
#pragma unroll
        for (int q = 0; q < 8; ++q) before += v[q];
    }
    for (; c < upto; ++c) before += chunk_sums[(size_t)c * SIZE + d];
    if (blockIdx.x == 0) { digit_total[d] = before; before = 0; }
    uint32_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = tile_counts[(size_t)(t + q) * SIZE + d];
#pragma unroll
        for (int q = 0; q < 8; ++q) { tile_counts[(size_t)(t + q) * SIZE + d] = before; before += v[q]; }
    }
    for (; t < t1; ++t) {
        const uint32_t v = tile_counts[(size_t)t * SIZE + d];
        tile_counts[(size_t)t * SIZE + d] = before;
        before += v;
    }
}

// ---- where a pass puts a pair --------------------------------------------------------------------------------------

// the sorted pairs, nothing else.  NARROW: fewer than 2^30 pairs of 4-byte keys, so a pair's place times four fits 32
// bits and a store is the array's address (the same in all lanes) plus a 32-bit offset per lane.
template <typename K>
struct pair_sink {
    K *kout; uint32_t *vout;
    template <bool NARROW>
    __device__ __forceinline__ void put(const uint32_t (&dest)[SORT_ITEMS], const K (&key)[SORT_ITEMS], const uint32_t (&val)[SORT_ITEMS],
                                        const bool (&live)[SORT_ITEMS]) const
    {
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            if (!live[i]) continue;
#ifdef PM_SORT_NOSTORE
            if (dest[i] != 0xfffffff0u) continue;       // (tuning: a pass without its stores)
#endif
            if constexpr (NARROW && sizeof(K) == 4) {
                const uint32_t off = dest[i] << 2;
                *(K *)((char *)kout + off) = key[i];
                *(uint32_t *)((char *)vout + off) = val[i];
            } else {
                kout[dest[i]] = key[i]; vout[dest[i]] = val[i];
            }
        }
    }
};

// ---- one pass ------------------------------------------------------------------------------------------------------
// Look-back state of (workgroup, digit): the top two bits say what the rest is -- nothing yet (0), the workgroup's
// own count of the digit (PARTIAL), or the count of the digit in this and all earlier workgroups (PREFIX).  S is
// 32-bit while the counts fit 30 bits, else 64-bit.
template <typename S> struct state_bits {
    static constexpr int SHIFT = 8 * sizeof(S) - 2;
    static constexpr S PARTIAL = (S)1 << SHIFT, PREFIX = (S)2 << SHIFT, COUNT = PARTIAL - 1;
};

template <typename K, unsigned RADIX> constexpr size_t pass_lds_bytes()
{
    constexpr size_t SIZE = (size_t)1 << RADIX;
    constexpr size_t hist = (size_t)SORT_WAVES * SIZE * sizeof(uint32_t), keys = (size_t)SORT_TILE * sizeof(K);
    return (hist > keys ? hist : keys) + (size_t)SORT_TILE * sizeof(uint32_t) + 2 * SIZE * sizeof(uint32_t) + 32 * sizeof(uint32_t);
}

// A workgroup takes the next tile of SORT_TILE pairs (tiles are handed out by a counter, so a tile's predecessors have
// all started), ranks its pairs by the digit at `bit` -- rank = place among the tile's pairs with that digit, in tile
// order --, learns from the look-back chain how many pairs with each digit the earlier tiles hold, brings its pairs
// into digit order in LDS and writes them out, each digit's pairs to consecutive addresses.
//
// Tile order is (wave, item, lane): wave w holds pairs w * 64 * ITEMS + i * 64 + lane.  A wave ranks one item at a
// time: the lanes whose digits agree find each other with one ballot per digit bit, read the wave's count of the
// digit, and the first of them adds their number to it.
// (two workgroups per CU: 8 waves per SIMD, which caps the kernel at 64 vector registers; it needs 47)
#ifndef PM_PASS_WAVES
#define PM_PASS_WAVES 8
#endif
#ifndef PM_SORT_LOOK
#define PM_SORT_LOOK 4
#endif
constexpr uint32_t LOOK = PM_SORT_LOOK;             // states of earlier tiles requested at a time
// ON_DEVICE: the number of pairs is *n_dev, known on the device only.  The grid then has a fixed, modest number of
// workgroups, each of which takes tiles until none is left -- so that a sort of few pairs (or of none) does not pay for
// the start of as many workgroups as the most pairs there could be would need.
// CHAINED = false: tile b is workgroup b's, `digit_base` holds the pairs with each digit in all tiles (k_scan_tiles'
// digit_total) and `states` is the scanned table of tile counts, of which the workgroup reads row b: no ticket, no
// look-back, nothing is published.
template <typename K, typename S, unsigned RADIX, bool COUNTED, bool ON_DEVICE, bool CHAINED, class Sink, bool STREAM = false>
__global__ void __launch_bounds__(SORT_BLOCK) __attribute__((amdgpu_waves_per_eu(PM_PASS_WAVES)))
k_radix_pass(const K *__restrict__ kin, const uint32_t *__restrict__ vin, uint32_t n, const uint32_t *__restrict__ digit_base,
             S *__restrict__ states, uint32_t *__restrict__ ticket, unsigned bit, uint32_t mask, Sink sink, const uint32_t *__restrict__ n_dev)
{
    static_assert(CHAINED || !ON_DEVICE, "a sort whose size the host does not know goes through the look-back chain");
    if constexpr (ON_DEVICE) {
        n = *n_dev;
        if ((uint64_t)blockIdx.x * SORT_TILE >= n) return;      // more workgroups than tiles: these take no ticket
    }
    constexpr uint32_t SIZE = 1u << RADIX;
    constexpr size_t HIST = (size_t)SORT_WAVES * SIZE * sizeof(uint32_t), KEYS = (size_t)SORT_TILE * sizeof(K);
    extern __shared__ __align__(16) unsigned char lds[];
    uint32_t *whist = (uint32_t *)lds;                  // per wave and digit: pairs seen so far, later the wave's first rank
    K *skeys = (K *)lds;                                // the same memory, once the ranks are known
    uint32_t *svals = (uint32_t *)(lds + (HIST > KEYS ? HIST : KEYS));
    uint32_t *dstart = svals + SORT_TILE;               // first place of a digit in the tile's digit order
    uint32_t *gdelta = dstart + SIZE;                   // place in the output minus place in the tile, per digit
    uint32_t *misc = gdelta + SIZE;                     // [0] the tile, [1..] scan scratch
    const int t = threadIdx.x, lane = lane_id(), w = wave_id();
    (void)lane;

  // (Measured and not kept: a ticket that is good for several consecutive tiles.  A workgroup then publishes the count of
  // its second tile only after its first is done, which waits for the second tile of the workgroup before it: the
  // pass becomes one chain, 8 ms instead of 90 us.)
  for (;;) {                                            // (one turn unless ON_DEVICE)
    if constexpr (CHAINED) { if (t == 0) misc[0] = atomicAdd(ticket, 1u); }
    for (uint32_t q = t; q < SORT_WAVES * SIZE; q += SORT_BLOCK) whist[q] = 0u;
    // (not chained: what the scan left for this tile and the digit totals are on their way while the pairs are ranked)
    uint32_t earlier_tiles = 0, digit_all = 0;
    if constexpr (!CHAINED) {
        if ((uint32_t)t < SIZE) {
            earlier_tiles = ((const uint32_t *)states)[(size_t)blockIdx.x * SIZE + t];
            digit_all = digit_base[t];
        }
    }
    __syncthreads();
    const uint32_t tile = CHAINED ? misc[0] : blockIdx.x;
    const uint32_t base = tile * (uint32_t)SORT_TILE;
    if (base >= n) return;                              // (ON_DEVICE: no tile is left; else never: the grid has exactly the tiles there are)
    const uint32_t count = n - base < (uint32_t)SORT_TILE ? n - base : (uint32_t)SORT_TILE;
    const uint32_t first = (uint32_t)w * (64u * SORT_ITEMS) + (uint32_t)lane;      // place in the tile of item 0

    K key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS], pos[SORT_ITEMS];
    const K *ktile = kin + base;                        // (a uniform base and a small index: no 64-bit address per lane)
    const uint32_t *vtile = COUNTED ? nullptr : vin + base;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const uint32_t q = first + 64u * i;
        const bool live = q < count;
        // (STREAM: a pass is the last reader of its input -- non-temporal loads, in sorts too large for the caches to matter)
        if constexpr (STREAM) {
            key[i] = live ? __builtin_nontemporal_load(ktile + q) : (K)0;
            val[i] = COUNTED ? base + q : (live ? __builtin_nontemporal_load(vtile + q) : 0u);
        } else {
            key[i] = live ? ktile[q] : (K)0;
            // COUNTED: the values that go in are the storage indices 0, 1, 2, ... -- nothing to read
            val[i] = COUNTED ? base + q : (live ? vtile[q] : 0u);
        }
    }

    // Rank of every pair among the pairs of its wave with the same digit.  The lanes whose digits agree find each
    // other one digit bit at a time: `mine` is the bit spread over a word, `set` the lanes that have it, and a lane
    // keeps the lanes that agree with it -- peers & ~(set ^ mine), one three-input bit operation per half of the wave.
    uint32_t *wh = whist + (size_t)w * SIZE;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const bool live = first + 64u * i < count;
        const uint32_t d = digit_of(key[i], bit, mask);
        const unsigned long long all = __ballot(live);
        uint32_t plo = (uint32_t)all, phi = (uint32_t)(all >> 32);
#pragma unroll
        for (unsigned b = 0; b < RADIX; ++b) {
            int32_t mine;                               // 0 or -1 (one instruction; the compiler spells a builtin as two shifts)
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mine) : "v"(d), "n"(b));
            const unsigned long long set = __ballot(mine != 0);
            plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)set, (uint32_t)mine, 0x90);
            phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(set >> 32), (uint32_t)mine, 0x90);
        }
        if (live) {
            const uint32_t seen = wh[d];                // every lane of the group reads the count ...
            const uint32_t before = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));     // peers in lower lanes
            pos[i] = seen + before;
            if (before == 0u) wh[d] = seen + (uint32_t)__popc(plo) + (uint32_t)__popc(phi);      // ... before its first lane raises it
        }
    }
    __syncthreads();

    // per digit: the waves' counts become the waves' first ranks; the tile's count goes to the look-back chain
    uint32_t total = 0;
    if ((uint32_t)t < SIZE) {
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < SORT_WAVES; ++q) {
            const uint32_t c = whist[(size_t)q * SIZE + t];
            whist[(size_t)q * SIZE + t] = run;
            run += c;
        }
        total = run;
        if constexpr (CHAINED)
            __hip_atomic_store(states + (size_t)tile * SIZE + t, (S)((tile == 0 ? state_bits<S>::PREFIX : state_bits<S>::PARTIAL) | (S)total),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t before = block_exclusive_sum(total, misc + 1);
    if constexpr (!CHAINED) {
        // first place of a digit in the output: the pairs with smaller digits in all tiles, then those with this digit in
        // the tiles before this one
        const uint32_t smaller = block_exclusive_sum(digit_all, misc + 1);
        if ((uint32_t)t < SIZE) {
            dstart[t] = before;
            gdelta[t] = smaller + earlier_tiles - before;
        }
    } else
    if ((uint32_t)t < SIZE) {
        uint32_t earlier = 0;                           // pairs with this digit in the tiles before this one
#ifdef PM_SORT_NOCHAIN
        if (false) {                                    // (tuning: a pass that does not wait for the tiles before it)
#else
        if (tile > 0) {
#endif
            // The states of the LOOK tiles before p are requested together: a request travels to the memory side and
            // back (the L2 of another XCD may hold the newer value), and a tile that started with many others finds its
            // nearest finished predecessor many tiles back.  They are then read in order, nearest first; at the first one
            // that has not counted yet the walk starts again from there.
            uint32_t p = tile - 1;
            for (bool done = false; !done;) {
                S s[LOOK];
#pragma unroll
                for (uint32_t k = 0; k < LOOK; ++k)
                    s[k] = k <= p ? __hip_atomic_load(states + (size_t)(p - k) * SIZE + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (S)0;
                uint32_t used = 0;
#pragma unroll
                for (uint32_t k = 0; k < LOOK; ++k) {
                    if (done || used != k) continue;    // (an earlier one of these was not ready, or the walk is over)
                    const unsigned flag = (unsigned)(s[k] >> state_bits<S>::SHIFT);
                    if (flag == 0u) continue;           // that tile has not counted yet (k <= p: tile 0 always has a PREFIX)
                    earlier += (uint32_t)(s[k] & state_bits<S>::COUNT);
                    used = k + 1;
                    if (flag == 2u) done = true;
                }
                p -= done ? 0u : used;
                if (used == 0u) __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(states + (size_t)tile * SIZE + t, (S)(state_bits<S>::PREFIX | (S)(earlier + total)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        dstart[t] = before;
        gdelta[t] = digit_base[t] + earlier - before;
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if (first + 64u * i < count) {
            const uint32_t d = digit_of(key[i], bit, mask);
            pos[i] += dstart[d] + wh[d];
        }
    }
    __syncthreads();                                    // the wave counts are not read any more: their memory takes the keys
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if (first + 64u * i < count) { skeys[pos[i]] = key[i]; svals[pos[i]] = val[i]; }
    }
    __syncthreads();

    uint32_t dest[SORT_ITEMS];
    bool live[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const uint32_t q = (uint32_t)i * SORT_BLOCK + (uint32_t)t;
        live[i] = q < count;
        key[i] = live[i] ? skeys[q] : (K)0;
        val[i] = live[i] ? svals[q] : 0u;
        dest[i] = gdelta[digit_of(key[i], bit, mask)] + q;
    }
    sink.template put<sizeof(S) == 4>(dest, key, val, live);
    if constexpr (!ON_DEVICE) return;
    __syncthreads();                                    // the tile's keys have been read out of LDS: the next tile may use it
  }
}

__global__ void k_clear(uint4 *__restrict__ p, size_t n)
{
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) p[q] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace

// Zeroes `bytes` bytes (a multiple of 16, 16-byte aligned) with a kernel of this library: the runtime's fill command does
// the same work but costs the stream 5 us more per sort.
hipError_t pm_launch_clear(void *p, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    if (bytes % 16 != 0 || ((uintptr_t)p & 15) != 0) return hipMemsetAsync(p, 0, bytes, s);
    const size_t n = bytes / 16, want = (n + 255) / 256;
    k_clear<<<(unsigned)(want < 2048 ? want : 2048), 256, 0, s>>>((uint4 *)p, n);
    return hipGetLastError();
}

namespace {

// ---- host side ---------------------------------------------------------------------------------------------------

constexpr size_t SORT_ALIGN = 256;
inline size_t aligned(size_t b) { return (b + SORT_ALIGN - 1) / SORT_ALIGN * SORT_ALIGN; }

// workspace: [ticket counters of every pass | digit counts of every pass | look-back states of every pass] (cleared by
// one fill per sort), then the scratch keys and values the passes alternate through
struct sort_plan {
    unsigned radix_bits, size, places, tiles;
    bool wide;                                          // 64-bit look-back states (2^30 pairs and more)
    size_t tickets, counts, states, cleared, keys_tmp, values_tmp, bytes;
};

sort_plan plan_sort(size_t n, unsigned bits, unsigned radix_bits, size_t key_bytes)
{
    sort_plan p{};
    p.radix_bits = radix_bits;
    p.size = 1u << radix_bits;
    p.places = (bits + radix_bits - 1) / radix_bits;
    p.tiles = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
    p.wide = n >= ((size_t)1 << 30);
    p.tickets = 0;
    p.counts = aligned(p.places * sizeof(uint32_t));
    p.states = p.counts + aligned((size_t)p.places * p.size * sizeof(uint32_t));
    // (look-back states of every place; or, for the other way, one table of tile counts and the sums of its chunks)
    const size_t chain_states = (size_t)p.places * p.size * p.tiles * (p.wide ? 8 : 4), table = ((size_t)p.tiles + SCAN_CHUNKS) * p.size * 4;
    p.cleared = p.states + aligned(chain_states > table ? chain_states : table);
    p.keys_tmp = p.cleared;
    p.values_tmp = p.keys_tmp + aligned(n * key_bytes);
    p.bytes = p.values_tmp + aligned(n * sizeof(uint32_t));
    return p;
}

// The fewest places wins: 10 bits per place where that saves a place over 9 and 8 (the 30 key bits of a 1024^3 mesh in
// three passes instead of four), else 9 where that saves one over 8 (27 bits of a 512^3 mesh: three passes), else 8.
unsigned place_bits(unsigned bits)
{
    if ((bits + 9) / 10 < (bits + 8) / 9 && (bits + 9) / 10 < (bits + 7) / 8) return 10;
    return (bits + 8) / 9 < (bits + 7) / 8 ? 9 : 8;
}

#ifndef PM_SORT_DEVICE_GRID
#define PM_SORT_DEVICE_GRID 512
#endif
template <typename K, typename S, unsigned RADIX, bool COUNTED, class Sink>
hipError_t launch_pass(const sort_plan &p, const K *kin, const uint32_t *vin, size_t n, const uint32_t *digit_base, void *states, uint32_t *ticket, unsigned bit,
                       unsigned now, Sink sink, int device, hipStream_t s, pm_ctx *timed, const uint32_t *n_dev, bool chained)
{
    constexpr size_t lds = pass_lds_bytes<K, RADIX>();
    auto allow_slot = [&](const void *kernel, int slot) -> hipError_t {
        if (lds <= 64 * 1024) return hipSuccess;
        // more LDS than a kernel gets by default: allowed once per kernel and device (a bit per device; setting it
        // twice from two threads is harmless)
        static std::atomic<uint64_t> allowed[4] = {};
        std::atomic<uint64_t> &a = allowed[slot];
        const uint64_t bit_of = 1ull << (device & 63);
        if (a.load(std::memory_order_acquire) & bit_of) return hipSuccess;
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) a.fetch_or(bit_of, std::memory_order_release);
        return e;
    };
    auto allow = [&](const void *kernel) { return allow_slot(kernel, n_dev ? 1 : chained ? 0 : 2); };
    auto allow_streaming = [&](const void *kernel) { return allow_slot(kernel, 3); };
    pm_timer timer(timed, PM_NSTAGES + PM_KERNEL_SORT_PASS);
    if (n_dev) {
        auto kernel = k_radix_pass<K, S, RADIX, COUNTED, true, true, Sink>;
        hipError_t e = allow((const void *)kernel);
        if (e != hipSuccess) return e;
        const unsigned grid = p.tiles < (unsigned)PM_SORT_DEVICE_GRID ? p.tiles : (unsigned)PM_SORT_DEVICE_GRID;
        kernel<<<grid, SORT_BLOCK, lds, s>>>(kin, vin, (uint32_t)n, digit_base, (S *)states, ticket, bit, (1u << now) - 1u, sink, n_dev);
    } else if (chained) {
        auto kernel = k_radix_pass<K, S, RADIX, COUNTED, false, true, Sink>;
        hipError_t e = allow((const void *)kernel);
        if (e != hipSuccess) return e;
        kernel<<<p.tiles, SORT_BLOCK, lds, s>>>(kin, vin, (uint32_t)n, digit_base, (S *)states, ticket, bit, (1u << now) - 1u, sink, n_dev);
    } else {
        // (a sort of more than 4M pairs of 4-byte keys reads its input as streaming data)
        if (sizeof(K) == 4 && n > ((size_t)1 << 22)) {
            auto kernel = k_radix_pass<K, S, RADIX, COUNTED, false, false, Sink, true>;
            hipError_t e = allow_streaming((const void *)kernel);
            if (e != hipSuccess) return e;
            kernel<<<p.tiles, SORT_BLOCK, lds, s>>>(kin, vin, (uint32_t)n, digit_base, (S *)states, ticket, bit, (1u << now) - 1u, sink, n_dev);
            return hipGetLastError();
        }
        auto kernel = k_radix_pass<K, S, RADIX, COUNTED, false, false, Sink>;
        hipError_t e = allow((const void *)kernel);
        if (e != hipSuccess) return e;
        kernel<<<p.tiles, SORT_BLOCK, lds, s>>>(kin, vin, (uint32_t)n, digit_base, (S *)states, ticket, bit, (1u << now) - 1u, sink, n_dev);
    }
    return hipGetLastError();
}

// Sorts of fewer tiles than this (one per CU: all of them start at once) keep the look-back chain: six launches a sort
// instead of twelve, and at 1M pairs the same time.  The environment variable
// PM_SORT_CHAIN_TILES sets another bound (tests: 0 sends every sort the other way, a huge value none).
unsigned chain_below_tiles()
{
    const char *v = std::getenv("PM_SORT_CHAIN_TILES");
    if (!v) return 256u;
    char *end = nullptr;
    const unsigned long long m = std::strtoull(v, &end, 10);
    return end != v ? (unsigned)(m > 0xffffffffull ? 0xffffffffull : m) : 256u;
}

// Sorts n (key, value) pairs by the low `bits` bits of the keys, values going in as 0, 1, 2, ... (vin == nullptr) or
// from vin; the sorted pairs end in kout / vout -- through `last`, the sink of the final pass.  n_dev != nullptr: the
// number of pairs is *n_dev <= n, read by the kernels (0 pairs: every kernel leaves at once).
template <typename K, typename S, unsigned RADIX, class LastSink>
hipError_t sort_pairs(void *tmp, size_t tmp_bytes, const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned bits, LastSink last,
                      int device, hipStream_t s, pm_ctx *timed, const uint32_t *n_dev)
{
    const sort_plan p = plan_sort(n, bits, RADIX, sizeof(K));
    if (p.bytes > tmp_bytes) return hipErrorOutOfMemory;
    char *w = (char *)tmp;
    uint32_t *tickets = (uint32_t *)(w + p.tickets), *counts = (uint32_t *)(w + p.counts);
    S *states = (S *)(w + p.states);
    K *ktmp = (K *)(w + p.keys_tmp);
    uint32_t *vtmp = (uint32_t *)(w + p.values_tmp);
    // Which way the tiles learn their places (see the top of the file).  The table of tile counts is written whole by the
    // counting kernel of each pass: nothing to clear.  A sort of a few tiles is all launches: it keeps the chain.
    const bool chained = n_dev != nullptr || p.tiles < chain_below_tiles() || sizeof(S) != 4;
    hipError_t e = hipSuccess;
    if (chained) {
        if ((e = pm_launch_clear(w, p.cleared, s)) != hipSuccess) return e;
        const size_t per_count = (size_t)PM_COUNT_BLOCK * COUNT_ITEMS;
        const size_t want = (n + per_count - 1) / per_count;
        k_digit_counts<K, RADIX><<<(unsigned)(want < PM_COUNT_GRID ? want : PM_COUNT_GRID), PM_COUNT_BLOCK, (size_t)p.places * p.size * sizeof(uint32_t), s>>>(
            kin, counts, (uint32_t)n, bits, n_dev);
        k_digit_offsets<RADIX><<<p.places, SORT_BLOCK, 0, s>>>(counts);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    // the passes alternate between the scratch and the output so that the last one ends in the output
    bool to_output = (p.places - 1) % 2 == 0, from_input = true;
    for (unsigned place = 0, bit = 0; place < p.places; ++place, bit += RADIX) {
        const K *ki = from_input ? kin : to_output ? ktmp : kout;
        const uint32_t *vi = from_input ? vin : to_output ? vtmp : vout;
        K *ko = to_output ? kout : ktmp;
        uint32_t *vo = to_output ? vout : vtmp;
        const unsigned now = bits - bit < RADIX ? bits - bit : RADIX;
        const uint32_t *base = counts + (size_t)place * p.size;
        S *st = states + (size_t)place * p.size * p.tiles;
        const bool counted = from_input && vin == nullptr;
        if (!chained) {
            // (the first place's part of the areas the chain would use: digit totals, the table of tile counts)
            base = counts; st = states;
            k_tile_counts<K, RADIX><<<p.tiles, SORT_BLOCK, 0, s>>>(ki, (uint32_t)n, (uint32_t *)st, bit, (1u << now) - 1u);
            // (the chunk sums live behind the table of tile counts, in the part of the area the chain's later places would use)
            const uint32_t per_chunk = (p.tiles + SCAN_CHUNKS - 1) / SCAN_CHUNKS < 8 ? 8 : (p.tiles + SCAN_CHUNKS - 1) / SCAN_CHUNKS;
            const uint32_t chunks = (p.tiles + per_chunk - 1) / per_chunk;
            uint32_t *chunk_sums = (uint32_t *)st + (size_t)p.tiles * p.size;
            k_chunk_sums<RADIX><<<chunks, p.size, 0, s>>>((const uint32_t *)st, p.tiles, per_chunk, chunk_sums);
            k_scan_chunks<RADIX><<<chunks, p.size, 0, s>>>((uint32_t *)st, p.tiles, per_chunk, chunk_sums, counts);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (place + 1 == p.places) {
            e = counted ? launch_pass<K, S, RADIX, true>(p, ki, vi, n, base, st, tickets + place, bit, now, last, device, s, timed, n_dev, chained)
                        : launch_pass<K, S, RADIX, false>(p, ki, vi, n, base, st, tickets + place, bit, now, last, device, s, timed, n_dev, chained);
        } else {
            const pair_sink<K> mid{ko, vo};
            e = counted ? launch_pass<K, S, RADIX, true>(p, ki, vi, n, base, st, tickets + place, bit, now, mid, device, s, timed, n_dev, chained)
                        : launch_pass<K, S, RADIX, false>(p, ki, vi, n, base, st, tickets + place, bit, now, mid, device, s, timed, n_dev, chained);
        }
        if (e != hipSuccess) return e;
        from_input = false;
        to_output = !to_output;
    }
    return hipSuccess;
}

template <typename K, class LastSink>
hipError_t sort_dispatch(void *tmp, size_t tmp_bytes, const K *kin, K *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned bits, LastSink last,
                         int device, hipStream_t s, pm_ctx *timed, const uint32_t *n_dev = nullptr)
{
    if (n == 0) return hipSuccess;
    if (bits < 1) bits = 1;
    const bool wide = n >= ((size_t)1 << 30);
    auto with_state = [&](auto radix) {
        constexpr unsigned R = decltype(radix)::value;
        return wide ? sort_pairs<K, uint64_t, R>(tmp, tmp_bytes, kin, kout, vin, vout, n, bits, last, device, s, timed, n_dev)
                    : sort_pairs<K, uint32_t, R>(tmp, tmp_bytes, kin, kout, vin, vout, n, bits, last, device, s, timed, n_dev);
    };
    if constexpr (sizeof(K) == 8) return with_state(std::integral_constant<unsigned, 8>{});
    else {
        switch (place_bits(bits)) {
        case 10: return with_state(std::integral_constant<unsigned, 10>{});
        case 9: return with_state(std::integral_constant<unsigned, 9>{});
        default: return with_state(std::integral_constant<unsigned, 8>{});
        }
    }
}

size_t tmp_bytes_for(size_t n, size_t key_bytes, unsigned max_bits)
{
    size_t most = 0;
    for (unsigned bits = 1; bits <= max_bits; ++bits) {
        const unsigned r = key_bytes == 8 ? 8 : place_bits(bits);
        const size_t b = plan_sort(n, bits, r, key_bytes).bytes;
        most = b > most ? b : most;
    }
    return most;
}

}  // namespace

size_t pm_sort_tmp_bytes(int64_t cap) { return tmp_bytes_for((size_t)cap, sizeof(uint32_t), 32); }

// key_in -> key_out, src: the cell keys in cell order and the storage slot each sorted entry came from
hipError_t pm_sort_pairs(pm_ctx *c, int bits)
{
    return sort_dispatch<uint32_t>(c->sort_tmp, c->sort_tmp_bytes, c->key_in, c->key_out, nullptr, c->src, (size_t)c->n, (unsigned)bits,
                                   pair_sink<uint32_t>{c->key_out, c->src}, c->device, c->stream, c);
}

// *n_dev <= n_max pairs of 32-bit keys, the count known on the device only: (kin, vin) -> (kout, vout); vin == nullptr: the
// values are 0, 1, 2, ...
size_t pm_sort_counted_tmp_bytes(int64_t n_max) { return tmp_bytes_for((size_t)n_max, sizeof(uint32_t), 32); }

hipError_t pm_sort_pairs_counted(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n_max,
                                 const uint32_t *n_dev, int bits, int device, hipStream_t s)
{
    return sort_dispatch<uint32_t>(tmp, tmp_bytes, kin, kout, vin, vout, n_max, (unsigned)bits, pair_sink<uint32_t>{kout, vout}, device, s, nullptr, n_dev);
}

// Particles leaving a slab: (destination slab, particle id) keys with the outbox slot as value.
size_t pm_sort_outbox_tmp_bytes(int64_t cap) { return tmp_bytes_for((size_t)cap, sizeof(uint64_t), 64); }

hipError_t pm_sort_outbox(pm_ctx *c, uint32_t count)
{
    // the keys are (slab << 32) | id: 32 bits of id and as many as the slab numbers take
    unsigned bits = 33;
    while (bits < 64 && ((uint64_t)1 << (bits - 32)) < (uint64_t)c->nslabs) ++bits;
    return sort_dispatch<uint64_t>(c->outbox_tmp, c->outbox_tmp_bytes, c->outbox_key, c->outbox_key2, nullptr, c->outbox_idx2, (size_t)count, bits,
                                   pair_sink<uint64_t>{c->outbox_key2, c->outbox_idx2}, c->device, c->stream, c);
}

hipError_t pm_sort_u64(void *tmp, size_t tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, uint32_t *order, size_t n, int device, hipStream_t s)
{
    return sort_dispatch<uint64_t>(tmp, tmp_bytes, keys_in, keys_out, nullptr, order, n, 64u, pair_sink<uint64_t>{keys_out, order}, device, s, nullptr);
}
