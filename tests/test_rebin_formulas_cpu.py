"""The arithmetic behind the binning that starts from the cell order of the step before (csrc/pm_rebin.hip), restated in
NumPy and held against what it must equal: the stable sort of the storage order by the new cell keys (functions.py:27-39
defines the cells; ties keep the storage order).  No GPU: this checks the formulas the kernels implement,
    stayer at slot r  ->  r - A(r) + B(r),        mover j (sorted by key, then slot)  ->  j + t_j - A(t_j),
with A(r) the movers stored before slot r, t_j the slot before which mover j goes in among the old slots, B(r) the
movers with t_j <= r -- for whole meshes and for a slab, where particles leave (the key past all cells) and arrive
(stored behind the old slots, no old key)."""
import numpy as np
import pytest


def merged_order(key_cur, key_new, rows, G):
    """key_cur: sorted keys of the n_old slots of the old order; key_new: keys of the n >= n_old slots now in use.
    Returns src: the slot each place of the new order takes its particle from."""
    n_old, n = len(key_cur), len(key_new)
    is_m = np.ones(n, bool)
    is_m[:n_old] = key_new[:n_old] != key_cur                     # arrivals (slots from n_old on) are movers one and all
    slots = np.nonzero(is_m)[0]
    mk = key_new[slots]
    o = np.argsort(mk, kind="stable")                             # the radix sort of the movers: by key, slots ascending
    mk, mi = mk[o], slots[o]
    m = len(mi)
    A = np.concatenate(([0], np.cumsum(is_m)))                    # A[t]: movers in slots < t, t = 0 .. n
    # where a mover goes in: inside the run of its new mesh row (row_start of the old order), before the cell's first
    # slot if it comes from below, behind its last if from above; behind everything if its key lies past all rows
    row_start = np.searchsorted(key_cur, np.arange(rows + 1) * G, "left")
    t = np.empty(m, np.int64)
    for j in range(m):
        row = mk[j] // G
        if row >= rows:
            t[j] = n_old
            continue
        lo, hi = row_start[row], row_start[row + 1]
        first = lo + np.searchsorted(key_cur[lo:hi], mk[j], "left")
        t[j] = first if mi[j] < first else lo + np.searchsorted(key_cur[lo:hi], mk[j], "right")
    assert np.all(np.diff(t) >= 0)                                # which is what lets B be one search per slot
    src = np.full(n, -1, np.int64)
    src[np.arange(m) + t - A[t]] = mi
    r = np.nonzero(~is_m)[0]
    src[r - A[r] + np.searchsorted(t, r, "right")] = r
    return src


@pytest.mark.parametrize("seed", range(40))
def test_whole_mesh(seed):
    rng = np.random.default_rng(seed)
    G = int(rng.integers(2, 6))
    rows, n = G * G, int(rng.integers(1, 500))
    key_cur = np.sort(rng.integers(0, rows * G, n))
    key_new = key_cur.copy()
    mv = rng.random(n) < rng.random()
    key_new[mv] = rng.integers(0, rows * G, mv.sum())             # anywhere: the box is periodic, a key may jump far
    assert np.array_equal(merged_order(key_cur, key_new, rows, G), np.argsort(key_new, kind="stable"))


@pytest.mark.parametrize("seed", range(40))
def test_slab_with_leavers_and_arrivals(seed):
    rng = np.random.default_rng(1000 + seed)
    G, Gs = 4, 2
    rows, last = Gs * G, Gs * G * G                               # `last`: the key of a particle whose cell is another slab's
    n_old, arrivals = int(rng.integers(1, 300)), int(rng.integers(0, 40))
    key_cur = np.sort(rng.integers(0, last, n_old))
    key_new = np.concatenate((key_cur, rng.integers(0, last, arrivals)))
    mv = rng.random(n_old) < 0.3
    key_new[:n_old][mv] = rng.integers(0, last + 1, mv.sum())     # some of them leave
    src = merged_order(key_cur, key_new, rows, G)
    assert np.array_equal(src, np.argsort(key_new, kind="stable"))
    assert np.all(key_new[src][len(src) - int((key_new == last).sum()):] == last)      # the leavers are the tail
