"""The margin of the thresholded KNN's first pass (include/hrag.h hrag_sim_topk_min_score, hipporag_amd/knn.py
_PREFIX_MARGIN), checked on the CPU: in the split layout of csrc/knn.hip split3_kernel a unit vector is x = hi + lo with
hi = fp16(x), lo = fp16(x - hi); the exact score is the chain hi.qhi + lo.qhi + hi.qlo and the first pass of
retrieve_knn(min_score=...) -- the reference's add_synonymy_edges reads neighbours down to synonymy_edge_sim_threshold only,
src/hipporag/HippoRAG.py:1004-1007 -- sees hi.qhi alone.  What it may miss by is |lo.qhi + hi.qlo| <= |lo| |qhi| + |hi| |qlo|
with |lo| <= 2^-11 |x| (+ the fp16 subnormal floor): 9.8e-4 for unit vectors.  No GPU involved."""
import numpy as np

from hipporag_amd.knn import _PREFIX_MARGIN


def _split(x):
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float16).astype(np.float32)
    return hi.astype(np.float64), lo.astype(np.float64)


def _unit(x):
    x = x.astype(np.float32)
    return x / np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), 1e-12).astype(np.float32)


def test_the_prefix_pass_misses_by_less_than_its_margin():
    rng = np.random.default_rng(5)
    worst, worst_lo = 0.0, 0.0
    for dim in (64, 128, 768, 1024, 4096):
        xs = [_unit(rng.standard_normal((400, dim))),
              _unit(rng.standard_normal((400, dim)) * rng.choice([1e-4, 1.0], (400, dim))),      # many tiny components
              _unit(np.eye(dim, dtype=np.float32)[:50] + 1e-3 * rng.standard_normal((50, dim))),  # nearly one-hot
              _unit(np.full((1, dim), 1.0) * (1 + 2.0 ** -11))]                                    # every component mid-way
        x = np.concatenate(xs)
        hi, lo = _split(x)
        nx = np.linalg.norm(hi + lo, axis=1)
        worst_lo = max(worst_lo, float((np.linalg.norm(lo, axis=1) / nx).max()))
        # queries: random ones, the keys themselves, and the adversarial direction for each key (aligned with its lo part)
        adv = _unit(lo.astype(np.float32) + 1e-30)
        for q in (_unit(rng.standard_normal((300, dim))), x[:300], adv[:300]):
            qhi, qlo = _split(q)
            rest = lo[: len(q)] @ qhi.T + hi[: len(q)] @ qlo.T            # [keys, queries]: the two dropped thirds
            worst = max(worst, float(np.abs(rest).max()))
            rest_self = np.einsum("ij,ij->i", lo[: len(q)], qhi) + np.einsum("ij,ij->i", hi[: len(q)], qlo)
            worst = max(worst, float(np.abs(rest_self).max()))
    assert worst_lo <= 2.0 ** -11 * 1.001 + 1e-6, worst_lo                 # |lo| <= 2^-11 |x|
    assert worst <= 2 * 2.0 ** -11 * 1.002 + 2e-6, worst                   # the bound of include/hrag.h: 9.8e-4
    assert worst < _PREFIX_MARGIN - 1e-4                                   # and the margin leaves room for the fp32 chain


def test_the_thresholded_selection_restated_returns_every_row_above_the_threshold():
    """The selection rule of hrag_sim_topk_min_score restated in numpy: tile maxima of the PREFIX scores, the (at most 16)
    tiles whose maximum reaches threshold - margin, exact scores for the rows of those tiles only, an overflow flag when
    a 17th tile reaches the cut.  With neighbours planted within 5e-3 of the threshold on both sides: for every query
    without the flag the rows found above the threshold are exactly ALL rows above it (so the caller's dense fallback is
    needed for the flagged queries only), and far fewer tiles are rescored than the 16 per query of the exact form."""
    rng = np.random.default_rng(3)
    dim, n_base, thr, tile = 128, 60, 0.8, 128
    base = _unit(rng.standard_normal((n_base, dim)))
    rows = [base]
    for i in range(n_base):
        n_dup = (2, 9, 30)[i % 3]
        u = _unit(rng.standard_normal((n_dup, dim)))
        u = _unit(u - (u @ base[i])[:, None] * base[i])
        c = rng.uniform(thr - 5e-3, thr + 5e-3, n_dup)[:, None].astype(np.float32)
        rows.append(c * base[i] + np.sqrt(1 - c * c) * u)
    keys = _unit(np.concatenate(rows + [_unit(rng.standard_normal((6000, dim)))]))
    perm = rng.permutation(len(keys))
    keys = keys[perm]
    q = keys[np.argsort(perm)[:n_base]]
    hi, lo = _split(keys)
    qhi, qlo = _split(q)
    prefix = hi @ qhi.T                                              # [keys, queries]
    exact = prefix + lo @ qhi.T + hi @ qlo.T
    n_tiles = -(-len(keys) // tile)
    pad = np.full((n_tiles * tile - len(keys), len(q)), -np.inf)
    tmax = np.concatenate([prefix, pad]).reshape(n_tiles, tile, len(q)).max(1)      # [tiles, queries]
    cut = thr - _PREFIX_MARGIN
    flagged, rescored = 0, 0
    for j in range(len(q)):
        order = np.argsort(-tmax[:, j], kind="stable")
        reach = order[tmax[order, j] >= cut]
        overflow = len(reach) > 16
        sel = reach[:16]
        rescored += len(sel)
        cand = np.concatenate([np.arange(t * tile, min((t + 1) * tile, len(keys))) for t in sel]) if len(sel) else np.empty(0, int)
        found = set(cand[exact[cand, j] >= thr].tolist())
        want = set(np.flatnonzero(exact[:, j] >= thr).tolist())
        if overflow:
            flagged += 1
            assert found <= want
        else:
            assert found == want, j
    assert 0 < flagged < len(q) // 2 and rescored < 16 * len(q) // 2
    near = np.abs(exact - thr) < _PREFIX_MARGIN
    assert (near & (exact >= thr)).any() and (near & (exact < thr)).any()    # the margin was exercised on both sides
