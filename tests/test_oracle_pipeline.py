"""Oracle pipeline known-answer tests (edge cases listed in SURVEY.md 8c)."""

import numpy as np
import pytest

import oracle
from tests.helpers import make_case


def test_min_max_normalize_cases():
    x = np.array([0.2, 0.5, -0.1], np.float32)
    y = oracle.min_max_normalize(x)
    assert y.dtype == np.float32 and y.min() == 0 and y.max() == 1
    np.testing.assert_array_equal(oracle.min_max_normalize(y), y)        # idempotent (HippoRAG.py:1627)
    np.testing.assert_array_equal(oracle.min_max_normalize(np.full(4, 0.3, np.float32)), np.ones(4, np.float32))


def test_tie_rule_is_score_desc_index_desc():
    x = np.array([1.0, 3.0, 3.0, 0.0, 3.0], np.float32)
    assert oracle.topk_desc(x).tolist() == [4, 2, 1, 0, 3]
    assert oracle.topk_desc(x, 2).tolist() == [4, 2]


def small_index():
    v = 10
    src = [0, 1, 2, 3, 6, 7, 8, 9]
    dst = [1, 2, 3, 0, 0, 1, 2, 3]
    a = oracle.build_symmetric_csr(v, src, dst, np.ones(len(src)))
    rng = np.random.default_rng(0)
    fe = rng.standard_normal((4, 8)).astype(np.float32)
    pe = rng.standard_normal((4, 8)).astype(np.float32)
    return oracle.RefIndex(fe, pe, np.array([0, 1, 2, 4]), np.array([1, 2, -1, 4]),
                           np.array([2, 1, 4, 1, 0, 0, 0, 0, 0, 0]), np.array([6, 7, 8, 9]),
                           oracle.column_normalize(a))


def test_seed_weights_mean_and_divisor():
    ix = small_index()
    scores = np.array([0.75, 0.25, 1.0, 0.5], np.float32)      # exactly representable
    ids, w = oracle.seed_weights(ix, scores, [0, 1, 2, 3])
    got = dict(zip(ids.tolist(), w.tolist()))
    # vertex 0: fact0 subj, 0.75/2                         -> 0.375
    # vertex 1: fact0 obj 0.75/1, fact1 subj 0.25/1        -> mean 0.5
    # vertex 2: fact1 obj 0.25/4, fact2 subj 1.0/4         -> mean 0.15625   (fact2's object is absent)
    # vertex 4: fact3 subj + obj (num_chunks 0: no division), twice 0.5 -> 0.5
    assert got == {0: 0.375, 1: 0.5, 2: 0.15625, 4: 0.5}
    assert ids.tolist() == [1, 4, 0, 2]    # tie 0.5 / 0.5 broken by first occurrence
    ids2, _ = oracle.seed_weights(ix, scores, [0, 1, 2, 3], link_top_k=2)
    assert ids2.tolist() == [1, 4]
    ids3, _ = oracle.seed_weights(ix, scores, [3, 0, 1, 2], link_top_k=2)   # filter order matters for ties
    assert ids3.tolist() == [4, 1]


def test_seed_assert_mirrors_reference():
    ix = small_index()
    scores = np.array([0.0, 0.4, 1.0, 0.6], np.float32)     # fact 0 has normalised score 0
    with pytest.raises(AssertionError):
        oracle.seed_weights(ix, scores, [0])


def test_retrieve_dpr_fallback_and_small_f():
    kg, pass_bits, fact_bits, index = make_case(400, 2400, 32, seed=3)
    rng = np.random.default_rng(1)
    qf = rng.standard_normal(32).astype(np.float32)
    qp = rng.standard_normal(32).astype(np.float32)
    res = oracle.retrieve_one(index, qf, qp, filter_fn=lambda cand: [])     # nothing survives
    ids, sc = oracle.retrieve_dpr_one(index, qp)
    assert res.used_dpr
    np.testing.assert_array_equal(res.sorted_doc_ids, ids)
    # F <= link_top_k: every fact is a candidate (HippoRAG.py:1683-1685)
    few = oracle.RefIndex(index.fact_emb[:3], index.passage_emb, index.subj_vertex[:3], index.obj_vertex[:3],
                          index.num_chunks, index.passage_vertex, index.p)
    s = oracle.fact_scores(few.fact_emb, qf)
    cand, kept = oracle.rerank_facts(s, 5)
    assert sorted(cand) == [0, 1, 2] and kept == cand
    # no facts at all -> empty scores -> DPR
    none = oracle.RefIndex(index.fact_emb[:0], index.passage_emb, index.subj_vertex[:0], index.obj_vertex[:0],
                           index.num_chunks, index.passage_vertex, index.p)
    assert oracle.retrieve_one(none, qf, qp).used_dpr


def test_power_mode_ranks_like_exact_mode():
    kg, pass_bits, fact_bits, index = make_case(3000, 30000, 64, seed=8)
    from hipporag_amd import synth
    from hipporag_amd.graph import bf16_bits_to_float
    qf = bf16_bits_to_float(synth.make_queries_np(fact_bits, 4, 1)[0])
    qp = bf16_bits_to_float(synth.make_queries_np(pass_bits, 4, 2)[0])
    for q in range(4):
        a = oracle.retrieve_one(index, qf[q], qp[q], ppr_mode="exact")
        b = oracle.retrieve_one(index, qf[q], qp[q], ppr_mode="power", ppr_iters=20)
        np.testing.assert_array_equal(a.sorted_doc_ids[:200], b.sorted_doc_ids[:200])
        np.testing.assert_allclose(b.sorted_doc_scores[:200], a.sorted_doc_scores[:200], rtol=1e-5)
        assert abs(a.x.sum() - 1) < 1e-12


def test_cpu_baseline_loop_agrees_with_oracle():
    from oracle.cpu_baseline import ReferenceStyleRetriever
    kg, pass_bits, fact_bits, index = make_case(1500, 12000, 48, seed=4)
    from hipporag_amd import synth
    from hipporag_amd.graph import bf16_bits_to_float
    qf = bf16_bits_to_float(synth.make_queries_np(fact_bits, 3, 1)[0])
    qp = bf16_bits_to_float(synth.make_queries_np(pass_bits, 3, 2)[0])
    ref = ReferenceStyleRetriever(index)
    for q in range(3):
        ids, sc = ref.retrieve_one(qf[q], qp[q])
        want = oracle.retrieve_one(index, qf[q], qp[q], exact_dot=False)
        np.testing.assert_array_equal(ids[:100], want.sorted_doc_ids[:100])
        np.testing.assert_allclose(sc[:100], want.sorted_doc_scores[:100], rtol=1e-7)
