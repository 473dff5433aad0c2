// Cell order of a step from the cell order of the step before, for sets in which few particles change their cell.
//
// What is wanted is the stable sort of the storage order by the new cell keys (functions.py:27-39 defines the cells;
// ties keep the storage order).  The storage order IS the sorted order of the old keys, so the particles whose key
// did not change (the stayers) are already in order among themselves, and only the others (the movers) have to be
// sorted: the result is the merge of the two sequences by (new key, storage index).  With A(r) the movers stored
// before slot r, j the place of a mover among the sorted movers, t_j the slot before which it goes in among the old
// slots (the first slot of its new cell if it comes from a cell below, the end of the cell if it comes from above), and
// B(r) the movers with t_j <= r,
//     a stayer at slot r goes to    r - A(r) + B(r),
//     mover j goes to               j + t_j - A(t_j).
// The drift itself marks the movers (a bit per slot); one kernel counts them per 64 slots (a look-back chain over tiles
// of the bit words) and compacts them, the radix sort of pm_sort.hip sorts them with their number known on the device only, one kernel places the
// movers, one the stayers.  No kernel needs the host to know how many movers there are, and the buffers hold as many as
// there are particles: whether this path or the sort of the whole set runs is a matter of speed only (the caller
// decides by an estimate), the permutation is the same.
#include "pm_internal.hpp"

namespace {

#ifndef PM_RB_ITEMS
#define PM_RB_ITEMS 8
#endif
constexpr int RB_BLOCK = 1024, RB_ITEMS = PM_RB_ITEMS, RB_TILE = RB_BLOCK * RB_ITEMS, RB_WAVES = RB_BLOCK / 64;
constexpr int RB_LIST = 2048;                           // inserted movers of a tile that are kept in LDS
constexpr uint32_t RB_PARTIAL = 1u << 30, RB_PREFIX = 2u << 30, RB_COUNT = RB_PARTIAL - 1u;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// first q in [lo, hi) with a[q] >= v (hi if none)
template <typename P>
__device__ __forceinline__ uint32_t lower_bound(P a, uint32_t lo, uint32_t hi, uint32_t v)
{
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The drift has left a bit per slot of the old order's n_old slots: which hold a mover.  A thread takes four words of 64
// slots, marks the slots from n_old on (a slab's arrivals, which have no place in the old order) and writes them back:
// mask[g] = which of slots 64 g .. 64 g + 63 hold a mover.  A tile is RS_TILE such words: before[g] = movers in the
// slots before 64 g, through a look-back chain over the tiles; and, if mk is given, (mk, mi) = the movers' new keys and
// slots, compacted in slot order (else k_compact_movers does that).  ctl (cleared by the caller): PM_RB_TICKET hands out the tiles; the last tile writes the number of
// movers.  states: one look-back word per tile, cleared by the caller.
constexpr int RS_ITEMS = 4, RS_TILE = RB_BLOCK * RS_ITEMS;

__global__ void __launch_bounds__(RB_BLOCK)
k_scan_movers(uint64_t *__restrict__ mask, uint32_t n_old, uint32_t n, uint32_t *__restrict__ before,
              uint32_t *__restrict__ states, uint32_t *__restrict__ ctl,
              const uint32_t *__restrict__ key_new, uint32_t *__restrict__ mk, uint32_t *__restrict__ mi)
{
    __shared__ uint32_t wsum[RB_WAVES], misc[2];
    const int t = threadIdx.x, lane = lane_id(), w = wave_id();
    if (t == 0) misc[0] = atomicAdd(ctl + PM_RB_TICKET, 1u);       // tiles in the order they start: a tile's predecessors are running
    __syncthreads();
    const uint32_t tile = misc[0], tiles = gridDim.x;
    const uint32_t groups = (n + 63u) >> 6;
    const uint32_t g0 = tile * (uint32_t)RS_TILE + (uint32_t)t * RS_ITEMS;      // this thread's first word
    uint64_t bits[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) bits[i] = g0 + i < groups ? mask[g0 + i] : 0ull;
    uint32_t mine = 0;
    uint32_t cnt[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint64_t first = (uint64_t)(g0 + i) << 6;            // the word's first slot
        // slots from n_old on are movers, slots from n on are no slots
        if (first + 64u > n_old) bits[i] |= first >= n_old ? ~0ull : ~0ull << (n_old - first);
        if (first + 64u > n) bits[i] &= first >= n ? 0ull : ~(~0ull << (n - first));
        if (g0 + i < groups) mask[g0 + i] = bits[i];
        cnt[i] = (uint32_t)__popcll(bits[i]);
        mine += cnt[i];
    }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const uint32_t in_wave = incl - mine;               // movers in the wave's words before this thread's
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t in_tile = 0, total = 0;                    // movers of the tile before this wave's words; of the whole tile
    for (int q = 0; q < RB_WAVES; ++q) { const uint32_t c = wsum[q]; if (q < w) in_tile += c; total += c; }
    if (w == 0) {
        // The first wave learns the movers of all tiles before this one: it reads the words of the 64 tiles before it at
        // once -- own count (PARTIAL) or count of the tile and everything before it (PREFIX) --, sums up to the nearest
        // PREFIX, and moves on by 64 tiles if there is none; a tile that has not counted yet is waited for (it has
        // started: tickets).  Before tile 0 there is nothing, which reads as a PREFIX of 0.
        if (lane == 0) __hip_atomic_store(states + tile, (tile == 0 ? RB_PREFIX : RB_PARTIAL) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t earlier = 0;
        if (tile > 0) {
            for (int64_t p = (int64_t)tile - 1;;) {
                const int64_t at = p - lane;
                const uint32_t s = at >= 0 ? __hip_atomic_load(states + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : RB_PREFIX;
                const uint32_t flag = s >> 30;
                const uint64_t ready = __ballot(flag != 0u), prefix = __ballot(flag == 2u);
                const int upto = prefix ? __ffsll((long long)prefix) - 1 : 63;      // lanes 0 .. upto are summed
                const uint64_t need = upto == 63 ? ~0ull : (1ull << (upto + 1)) - 1ull;
                if ((ready & need) != need) { __builtin_amdgcn_s_sleep(1); continue; }
                uint32_t v = lane <= upto ? (s & RB_COUNT) : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                earlier += v;
                if (prefix) break;
                p -= 64;
            }
            if (lane == 0) __hip_atomic_store(states + tile, RB_PREFIX | (earlier + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            misc[1] = earlier;
            if (tile + 1 == tiles) {
                const uint32_t m = earlier + total;
                ctl[PM_RB_MOVERS] = m;
                // one entry past the last word, for a mover that goes in behind the last slot
                mask[groups] = 0ull;
                before[groups] = m;
            }
        }
    }
    __syncthreads();
    uint32_t at = misc[1] + in_tile + in_wave;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const uint32_t g = g0 + (uint32_t)i;
        if (g >= groups) break;
        before[g] = at;
        if (mk == nullptr) { at += cnt[i]; continue; }
        // few movers: their keys and slots are compacted here, a turn per set bit (mostly no turn at all)
        for (uint64_t b = bits[i]; b != 0ull; b &= b - 1ull, ++at) {
            const uint32_t slot = (g << 6) + (uint32_t)(__ffsll((long long)b) - 1);
            mk[at] = key_new[slot];
            mi[at] = slot;
        }
    }
}

// The movers' new keys and slots, compacted in slot order: a thread per slot.  (The loop over the set bits of a word in
// the kernel above does the same 20 us sooner when one particle in a hundred moves, and 60 us later when six do: the
// caller chooses by the estimate it has.)
__global__ void __launch_bounds__(256)
k_compact_movers(const uint64_t *__restrict__ mask, const uint32_t *__restrict__ before, const uint32_t *__restrict__ key_new, uint32_t n,
                 uint32_t *__restrict__ mk, uint32_t *__restrict__ mi)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint64_t word = mask[r >> 6];
    if (!((word >> (r & 63u)) & 1ull)) return;
    const uint32_t at = before[r >> 6] + (uint32_t)__popcll(word & ((1ull << (r & 63u)) - 1ull));
    mk[at] = key_new[r];
    mi[at] = r;
}

// movers stored before slot t (t <= n)
__device__ __forceinline__ uint32_t movers_before(const uint64_t *__restrict__ mask, const uint32_t *__restrict__ before, uint32_t t)
{
    const uint32_t g = t >> 6;
    return before[g] + (uint32_t)__popcll(mask[g] & ((1ull << (t & 63u)) - 1ull));
}

// Where sorted mover (k, slot) goes in among the old slots.  The cells of a mesh row are one run of slots (row_start,
// as the last binning left it), so the search for the new cell's slots stays inside that run.
// rows, n_old: the mesh rows and the slots of the old order.  A key past all rows (a particle that has left a slab) goes
// in behind everything; a slot past n_old (a particle that has just arrived in one) counts as coming from above.
__device__ __forceinline__ uint32_t goes_in_at(uint32_t k, uint32_t slot, const uint32_t *__restrict__ key_cur, const uint32_t *__restrict__ row_start, uint32_t G,
                                               uint32_t rows, uint32_t n_old)
{
    const uint32_t row = k / G;
    if (row >= rows) return n_old;
    const uint32_t lo = row_start[row], hi = row_start[row + 1];
    const uint32_t first = lower_bound(key_cur, lo, hi, k);
    // from a cell below: before the cell's first slot; from a cell above: behind its last one
    return slot < first ? first : lower_bound(key_cur, first, hi, k + 1u);
}

// Sorted mover j: the slot t_j before which it goes in (kept in `at`), and its place.
__global__ void __launch_bounds__(256)
k_place_movers(const uint32_t *__restrict__ mk, const uint32_t *__restrict__ mi, const uint32_t *__restrict__ ctl,
               const uint32_t *__restrict__ key_cur, const uint32_t *__restrict__ row_start, uint32_t G,
               const uint64_t *__restrict__ mask, const uint32_t *__restrict__ before,
               uint32_t *__restrict__ src, uint32_t *__restrict__ key_sorted, uint32_t *__restrict__ at, uint32_t rows, uint32_t n_old)
{
    const uint32_t m = ctl[PM_RB_MOVERS];
    for (uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x; j < m; j += (uint64_t)gridDim.x * 256u) {
        const uint32_t k = mk[j], slot = mi[j];
        const uint32_t t = goes_in_at(k, slot, key_cur, row_start, G, rows, n_old);
        const uint32_t pos = (uint32_t)j + t - movers_before(mask, before, t);
        src[pos] = slot;
        key_sorted[pos] = k;
        at[j] = t;
    }
}

// tile_first[T] = the movers that go in before the first slot of tile T of the slots, T = 0 .. tiles (`at` is sorted: the
// movers are)
__global__ void __launch_bounds__(256)
k_tile_first(const uint32_t *__restrict__ at, const uint32_t *__restrict__ ctl, uint32_t *__restrict__ tile_first, uint32_t tiles)
{
    const uint32_t T = blockIdx.x * 256u + threadIdx.x;
    if (T <= tiles) tile_first[T] = lower_bound(at, 0u, ctl[PM_RB_MOVERS], T * (uint32_t)RB_TILE);
}

// Where the mesh rows begin in the new order, from where they began in the old one: the particles with a key below the
// row's first are the stayers among the old slots before the row's old beginning, and the movers whose new key is
// below it.  In place (a thread reads and writes its own entry); rows = entries - 1.  (Bisecting the new keys, as after
// a sort of everything, takes 33 us on a uniform set of 16.7 M particles; this takes a quarter of it.)
__global__ void __launch_bounds__(256)
k_row_start_update(uint32_t *__restrict__ row_start, uint32_t rows, uint32_t G, const uint64_t *__restrict__ mask, const uint32_t *__restrict__ before,
                   const uint32_t *__restrict__ mk, const uint32_t *__restrict__ ctl)
{
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row > rows) return;
    const uint32_t old = row_start[row], m = ctl[PM_RB_MOVERS];
    const uint64_t first_key = (uint64_t)row * G;
    uint32_t lo = 0, hi = m;                            // movers with a key below first_key
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)mk[mid] < first_key) lo = mid + 1; else hi = mid;
    }
    row_start[row] = old - movers_before(mask, before, old) + lo;
}

// The stayers of a tile of slots: r - A(r) + B(r).  The movers that go in inside the tile (tile_first says which) are
// kept in LDS, where every group of 64 slots finds its own.
__global__ void __launch_bounds__(RB_BLOCK)
k_place_stayers(const uint32_t *__restrict__ key_new, uint32_t n, const uint64_t *__restrict__ mask, const uint32_t *__restrict__ before,
                const uint32_t *__restrict__ at, const uint32_t *__restrict__ tile_first, uint32_t *__restrict__ src, uint32_t *__restrict__ key_sorted)
{
    __shared__ uint32_t list[RB_LIST], gfirst[RB_TILE / 64 + 1];
    const int t = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t base = blockIdx.x * (uint32_t)RB_TILE;
    const uint32_t b0 = tile_first[blockIdx.x], cnt = tile_first[blockIdx.x + 1] - b0;     // movers with at < base; with base <= at < base + RB_TILE
    const bool in_lds = cnt <= (uint32_t)RB_LIST;
    if (in_lds) {
        for (uint32_t q = t; q < cnt; q += RB_BLOCK) list[q] = at[b0 + q];
        __syncthreads();
        // gfirst[g] = movers (of the tile's list) that go in before the first slot of the tile's group g
        if (t <= RB_TILE / 64) gfirst[t] = lower_bound(list, 0u, cnt, base + 64u * (uint32_t)t);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RB_ITEMS; ++i) {
        const uint32_t gl = (uint32_t)w * RB_ITEMS + (uint32_t)i;       // group of the tile
        const uint32_t r = base + 64u * gl + (uint32_t)lane;
        if (base + 64u * gl >= n) break;
        const uint64_t movers = mask[r >> 6];
        const uint32_t a = before[r >> 6] + (uint32_t)__popcll(movers & ((1ull << lane) - 1ull));
        uint32_t b;
        if (in_lds) {
            const uint32_t f = gfirst[gl], l = gfirst[gl + 1];
            b = f;
            for (uint32_t q = f; q < l; ++q) b += list[q] <= r ? 1u : 0u;       // (mostly none: f == l)
            b += b0;
        } else {
            b = lower_bound(at, b0, b0 + cnt, r + 1u);
        }
        if (r < n && !((movers >> lane) & 1ull)) {
            const uint32_t pos = r - a + b;
            src[pos] = r;
            key_sorted[pos] = key_new[r];
        }
    }
}

}  // namespace

size_t pm_rebin_ctl_bytes(int64_t cap)
{
    const size_t tiles = ((size_t)cap + RB_TILE - 1) / RB_TILE;
    return (PM_RB_WORDS + 2 * tiles + 2 + 4) * sizeof(uint32_t);     // control words, look-back words, tile_first (cleared in 16-byte pieces)
}

// key_out (the sorted keys of the old order's order_n slots), key_in (the new keys of the n slots now in use: the same
// slots, and behind them the arrivals of a slab), rb_mask (where the two differ: written by the drift), row_start (of the
// old order) -> src and key_alt, as pm_sort_pairs would leave src and key_out, and row_start of the new order
// (pm_launch_row_start need not run).  Returns with key_out and key_alt exchanged.
hipError_t pm_rebin_incremental(pm_ctx *c, int bits)
{
    const uint32_t n = (uint32_t)c->n;
    if (n == 0) return hipSuccess;
    const uint32_t tiles = (n + RB_TILE - 1) / RB_TILE;                 // of slots (k_place_stayers)
    const uint32_t groups = (n + 63u) >> 6, stiles = (groups + RS_TILE - 1) / RS_TILE;      // of mask words (k_scan_movers)
    hipError_t e = pm_launch_clear(c->rb_ctl, ((PM_RB_WORDS + (size_t)stiles + tiles + 1) * sizeof(uint32_t) + 15) / 16 * 16, c->stream);
    if (e != hipSuccess) return e;
    uint32_t *ctl = c->rb_ctl, *states = c->rb_ctl + PM_RB_WORDS, *tile_first = states + stiles;
    // (the estimate is a guide to speed, as everywhere: either way compacts the same movers)
    const bool very_few = c->movers_seen && (int64_t)*c->movers_seen * 64 <= c->n;
    k_scan_movers<<<stiles, RB_BLOCK, 0, c->stream>>>(c->rb_mask, (uint32_t)c->order_n, n, c->rb_before, states, ctl, c->key_in,
                                                      very_few ? c->rb_key[0] : nullptr, c->rb_slot[0]);
    if (!very_few) k_compact_movers<<<(n + 255u) / 256u, 256, 0, c->stream>>>(c->rb_mask, c->rb_before, c->key_in, n, c->rb_key[0], c->rb_slot[0]);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // (the workspace of the sort of the whole set, which does not run in a binning that takes this path)
    e = pm_sort_pairs_counted(c->sort_tmp, c->sort_tmp_bytes, c->rb_key[0], c->rb_key[1], c->rb_slot[0], c->rb_slot[1], n, ctl + PM_RB_MOVERS, bits,
                              c->device, c->stream);
    if (e != hipSuccess) return e;
    k_place_movers<<<1024, 256, 0, c->stream>>>(c->rb_key[1], c->rb_slot[1], ctl, c->key_out, c->row_start, (uint32_t)c->g.G, c->rb_mask, c->rb_before, c->src,
                                                c->key_alt, c->rb_at, (uint32_t)c->g.Gs * (uint32_t)c->g.G, (uint32_t)c->order_n);
    k_tile_first<<<(tiles + 1u + 255u) / 256u, 256, 0, c->stream>>>(c->rb_at, ctl, tile_first, tiles);
    k_place_stayers<<<tiles, RB_BLOCK, 0, c->stream>>>(c->key_in, n, c->rb_mask, c->rb_before, c->rb_at, tile_first, c->src, c->key_alt);
    const uint32_t rows = (uint32_t)c->g.Gs * (uint32_t)c->g.G;
    k_row_start_update<<<(rows + 1u + 255u) / 256u, 256, 0, c->stream>>>(c->row_start, rows, (uint32_t)c->g.G, c->rb_mask, c->rb_before, c->rb_key[1], ctl);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    uint32_t *k = c->key_out; c->key_out = c->key_alt; c->key_alt = k;
    return hipSuccess;
}
