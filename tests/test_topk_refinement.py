"""The bound refinement of csrc/topk.hip row_topk_kernel (round 6: 1024-bin radix histograms instead of bisecting the
64-bit key space), restated in numpy and run on the distributions that defeat a threshold taken from thread-local maxima:
all-equal scores (keys differ in the index bits only), heavy ties, a dense cluster just above the bound, huge dynamic
range.  Checks the loop's invariant and exit condition -- kk <= count(key >= T) <= capacity -- and that it terminates in
a handful of passes; the kernel itself is checked against np.argsort on the GPU (tests/test_gpu_parity.py
test_topk_rows_exact, 875 000 x 2047).  Replaces np.argsort(...)[::-1] of reference src/hipporag/HippoRAG.py:1500,1688,1746
and torch.topk of utils/embed_utils.py:55,73 wherever only a prefix is read."""
import numpy as np
import pytest

CAP, BINS = 4096, 1024


def _ordered(f32):
    u = f32.view(np.uint32).astype(np.uint64)
    return np.where(u >> np.uint64(31), ~u & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))


def _keys(scores):
    return (_ordered(scores.astype(np.float32)) << np.uint64(32)) | np.arange(len(scores), dtype=np.uint64)


def refine(keys, kk, lo):
    """-> (T, passes): the loop of row_topk_kernel's overflow path, one histogram pass per iteration"""
    hi = int(keys.max())
    lo, above, passes = int(lo), 0, 0
    while True:
        passes += 1
        width = hi - lo
        shift = 0
        while (width >> shift) >= BINS:
            shift += 1
        inside = keys[(keys >= np.uint64(lo)) & (keys <= np.uint64(hi))]
        hist = np.bincount(((inside - np.uint64(lo)) >> np.uint64(shift)).astype(np.int64), minlength=BINS)
        suffix = np.cumsum(hist[::-1])[::-1]                        # keys in the bins >= b
        assert above < kk <= above + suffix[0]                      # the invariant
        b = int(np.flatnonzero(above + suffix >= kk).max())
        c_ge = above + int(suffix[b])
        c_gt = above + (int(suffix[b + 1]) if b + 1 < BINS else 0)
        blo = lo + (b << shift)
        if c_ge <= CAP or shift == 0:
            return blo, passes
        above, lo, hi = c_gt, blo, min(hi, blo + (1 << shift) - 1)


@pytest.mark.parametrize("name", ["normal", "all_equal", "heavy_ties", "cluster_above_the_bound", "wide_range", "two_values"])
@pytest.mark.parametrize("kk", [1, 200, 2047, 2048])
def test_refinement_reaches_a_bound_that_fits(name, kk):
    rng = np.random.default_rng(len(name) + kk)
    n = 300_000
    s = {"normal": lambda: rng.standard_normal(n),
         "all_equal": lambda: np.full(n, 0.25),
         "heavy_ties": lambda: np.round(rng.standard_normal(n) * 4) / 4,
         "cluster_above_the_bound": lambda: np.where(rng.random(n) < 0.2, 0.5 + 1e-6 * rng.random(n), rng.random(n) * 0.4),
         "wide_range": lambda: rng.standard_normal(n) * 10.0 ** rng.integers(-30, 30, n),
         "two_values": lambda: rng.choice([-1.0, 1.0], n)}[name]().astype(np.float32)
    keys = _keys(s)
    start = np.sort(keys)[max(0, n - 20000)]                        # a loose bound: 20 000 keys at or above it
    T, passes = refine(keys, kk, start)
    count = int((keys >= np.uint64(T)).sum())
    assert kk <= count <= CAP, (name, kk, count)
    assert passes <= 7                                              # 64 bits at 10 bits per pass
    want = np.sort(keys)[::-1][:kk]                                 # and the top-kk of the candidates are THE top-kk
    got = np.sort(keys[keys >= np.uint64(T)])[::-1][:kk]
    np.testing.assert_array_equal(got, want)
