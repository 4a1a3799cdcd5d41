"""Parity checkers shared by tests/, __graft_entry__.smoke() and bench.py's parity spot check.  TEST INFRASTRUCTURE
(like the rest of oracle/): the product never imports this.  What "identical ranked ids" means where the reference
leaves the order undefined (ties), in two windows -- the contract's 4 ulp(fp32) and the one that follows
the measured score error -- and the score bars."""

from __future__ import annotations

import numpy as np


def make_case(num_vertices: int, num_edges: int, dim: int, seed: int, passage_frac: float = 0.125,
              power_law: bool = False):
    """Returns (kg, pass_bits, fact_bits, RefIndex) -- the same data for engine and oracle."""
    import oracle
    from hipporag_amd import synth
    from hipporag_amd.graph import bf16_bits_to_float
    kg = synth.make_kg(num_vertices, num_edges, seed, passage_frac=passage_frac, power_law=power_law)
    pass_bits = synth.make_embeddings_np(kg.n_passages, dim, seed + 1)
    fact_bits = synth.make_embeddings_np(kg.n_facts, dim, seed + 2)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    p = oracle.column_normalize(a)
    index = oracle.RefIndex(fact_emb=bf16_bits_to_float(fact_bits), passage_emb=bf16_bits_to_float(pass_bits),
                            subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex,
                            num_chunks=kg.num_chunks, passage_vertex=kg.passage_vertex, p=p)
    return kg, pass_bits, fact_bits, index



def tie_aware_equal(got_ids, ref_ids, ref_scores, rel_gap=4e-6, abs_gap=0.0):
    """Ranked ids equal, except inside runs of reference scores closer than the gap (tie classes,
    SURVEY.md 8c): there only set-equality is required; the class cut by the top-k boundary is
    not checked (its members may legitimately come from just beyond the boundary)."""
    got_ids = np.asarray(got_ids)
    ref_ids = np.asarray(ref_ids)
    if got_ids.shape != ref_ids.shape:
        return False
    if np.array_equal(got_ids, ref_ids):
        return True
    s = np.asarray(ref_scores, dtype=np.float64)
    n = len(ref_ids)
    start = 0
    while start < n:
        end = start + 1
        while end < n and abs(s[end - 1] - s[end]) <= max(abs_gap, rel_gap * abs(s[end - 1])):
            end += 1
        if end < n and set(got_ids[start:end].tolist()) != set(ref_ids[start:end].tolist()):
            return False
        start = end
    return True


def tie_aware_report(got_ids, ref_ids, ref_scores, rel_gap=4e-6, abs_gap=0.0):
    """tie_aware_equal plus HOW MUCH of the agreement is exact: {"equal": the tie-class-aware verdict,
    "exact_positions": ranks at which the two id lists agree outright, "n": ranks compared}.  The ranks that
    only agree as members of a tie class are n - exact_positions."""
    got = np.asarray(got_ids)
    ref = np.asarray(ref_ids)
    same = int((got == ref).sum()) if got.shape == ref.shape else 0
    return {"equal": bool(tie_aware_equal(got, ref, ref_scores, rel_gap=rel_gap, abs_gap=abs_gap)),
            "exact_positions": same, "n": int(ref.size)}


ID_GAP_FLOOR = 2e-6   # ten times the relative score error the device shows at the BASELINE sizes (5.5e-7 at cfg 3)


def ranked_parity(got_ids, got_scores, ref_sorted_ids, ref_sorted_scores, ref_full_scores, gap_cap=2e-5):
    """Score + ranking parity of ONE query against the oracle, with a tie window that follows the MEASURED error.

    worst = max relative deviation of the returned scores from the oracle's score of the same passage.  Two passages
    can only come back in the other order when their oracle scores are closer than 2 * worst relative (each side is
    off by at most worst), so that -- 2.2 * worst, never below ID_GAP_FLOOR, never above the historical 2e-5 -- is the
    tie window inside which a permutation is accepted; everywhere else the ids must be identical.  Returns
    {"equal", "worst_rel_err", "rel_gap", "exact_positions", "n"}."""
    got_ids = np.asarray(got_ids)
    want = np.asarray(ref_full_scores, dtype=np.float64)[got_ids]
    got = np.asarray(got_scores, dtype=np.float64)
    nz = want > 0
    worst = float(np.abs(got[nz] / want[nz] - 1).max()) if nz.any() else 0.0
    zeros_ok = bool(np.all(got[~nz] == 0))
    gap = min(max(ID_GAP_FLOOR, 2.2 * worst), gap_cap)
    k = len(got_ids)
    rep = tie_aware_report(got_ids, np.asarray(ref_sorted_ids)[:k], np.asarray(ref_sorted_scores)[:k], rel_gap=gap)
    rep.update(worst_rel_err=worst, rel_gap=gap, equal=bool(rep["equal"] and zeros_ok))
    return rep


def prior_noise_allowance(index, q_pass) -> np.ndarray:
    """Per-passage relative allowance for what the REFERENCE leaves undefined: its passage prior is
    min_max_normalize(np.dot(passage_embeddings, q)) in fp32 (HippoRAG.py:1496-1498, misc_utils.py:130-139), and the
    fp32 dot product carries ~1e-7 of summation-order noise (two BLAS builds differ by it).  A passage whose
    normalised score is s takes that noise into its prior -- and, on graphs where a passage's PPR score is dominated
    by its own prior (the ring: passages 8 hops apart), into its final score -- at ~1e-7 / (range * s) relative: the
    oracle's own two dot variants (exact fp64 vs fp32 BLAS) differ by 1.5e-5 at a passage with s = 0.0016.  Returned:
    a model floor (6e-8 / range) / s for every passage, in passage order; callers add it to the score bar."""
    from oracle.hipporag_ref import _dot
    raw = _dot(index.passage_emb, np.asarray(q_pass, dtype=np.float32), True).astype(np.float64)
    rng = float(raw.max() - raw.min())
    if rng <= 0:
        return np.zeros(len(raw))
    nrm = (raw - raw.min()) / rng
    return (6e-8 / rng) / np.maximum(nrm, 1e-9)


ULP4_FP32 = 4 * 2.0 ** -23     # set-equality only where adjacent oracle scores differ by <= 4 ulp(fp32)


def ulp4_report(got_ids, ref_sorted_ids, ref_sorted_scores):
    """The id verdict at SURVEY 8(c)'s own tie window (4 ulp of fp32 = 4.8e-7 relative) -- the contract's window, next to
    the measured-error window of ranked_parity; {"equal", "exact_positions", "n"}."""
    k = len(got_ids)
    return tie_aware_report(got_ids, np.asarray(ref_sorted_ids)[:k], np.asarray(ref_sorted_scores)[:k], rel_gap=ULP4_FP32)


def percentiles(values, qs=(50, 99, 100)):
    """{"p50": ..., "p99": ..., "max": ...} of a list of non-negative errors (empty -> None)."""
    v = np.asarray(list(values), dtype=np.float64)
    if v.size == 0:
        return None
    out = {}
    for q in qs:
        out["max" if q == 100 else f"p{q}"] = float(np.percentile(v, q))
    return out
