"""The index arithmetic of the grouped pass 3 (csrc/sim_gemm.hip: pair_hist_kernel, pair_scan_kernel, pair_scatter_kernel,
tile_rescore_grouped_kernel -- the rescore of the fused top-k that replaces torch.topk over a materialised score block,
reference src/hipporag/utils/embed_utils.py:53-73) restated in numpy: bucket the (tile, query) pairs by tile, cut every
bucket into chunks of 16, let workgroup w find its chunk by the binary search the kernel uses.  Every valid pair must be
covered exactly once, by a chunk of its own tile; "no tile" entries take no part; the launch grid is an upper bound."""
import numpy as np
import pytest

CHUNK = 16


def _workgroup(w, chunk_start, bucket_end, n_tiles):
    lo, hi = 0, n_tiles                       # chunk_start[lo] <= w < chunk_start[hi]
    while hi - lo > 1:
        mid = (lo + hi) >> 1
        if chunk_start[mid] <= w:
            lo = mid
        else:
            hi = mid
    t = lo
    first = (0 if t == 0 else bucket_end[t - 1]) + (w - chunk_start[t]) * CHUNK
    return t, first, min(CHUNK, bucket_end[t] - first)


@pytest.mark.parametrize("n_tiles,batch,k,p_valid,skew", [(64, 300, 16, 1.0, 0.0), (6836, 4096, 16, 0.08, 0.0), (157, 1024, 16, 1.0, 2.0),
                                                         (100, 64, 5, 0.5, 1.0), (64, 17, 16, 0.0, 0.0), (3, 40, 2, 1.0, 0.0)])
def test_every_valid_pair_is_rescored_exactly_once_by_a_chunk_of_its_tile(n_tiles, batch, k, p_valid, skew):
    rng = np.random.default_rng(n_tiles + batch)
    w_t = (1.0 / (1 + np.arange(n_tiles))) ** skew
    sel = rng.choice(n_tiles, size=(batch, k), p=w_t / w_t.sum()).astype(np.int64)
    n_valid = (rng.random(batch) < p_valid) * rng.integers(1, k + 1, batch)        # the selected tiles are a PREFIX of a record
    sel[np.arange(k)[None, :] >= n_valid[:, None]] = -1
    pairs = np.arange(batch * k)
    tile_of = sel.reshape(-1)
    valid = tile_of >= 0
    hist = np.bincount(tile_of[valid], minlength=n_tiles)                            # pair_hist_kernel
    cursor = np.concatenate([[0], np.cumsum(hist)[:-1]])                             # pair_scan_kernel: first position ...
    n_chunks = -(-hist // CHUNK)
    chunk_start = np.concatenate([[0], np.cumsum(n_chunks)])                         # ... and chunks before bucket t; [n] = all
    order = np.empty(int(valid.sum()), np.int64)
    end = cursor.copy()
    for p in rng.permutation(pairs[valid]):                                          # pair_scatter_kernel: any arrival order
        order[end[tile_of[p]]] = p
        end[tile_of[p]] += 1
    grid = -(-batch * k // CHUNK) + n_tiles                                          # the launch's upper bound
    total = int(chunk_start[n_tiles])
    assert total <= grid
    seen = np.zeros(batch * k, np.int64)
    for w in range(total):
        t, first, n = _workgroup(w, chunk_start, end, n_tiles)
        assert 1 <= n <= CHUNK
        mine = order[first: first + n]
        assert np.all(tile_of[mine] == t)
        seen[mine] += 1
    assert np.all(seen[valid] == 1) and np.all(seen[~valid] == 0)
    # the arrival counter of a query waits for its valid tiles: one arrival per valid pair
    arrivals = np.bincount(pairs[valid] // k, minlength=batch)
    assert np.array_equal(arrivals, (sel >= 0).sum(1))
