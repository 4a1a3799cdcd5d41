"""Host-side logic of the path that needs no GPU: the graph compiler's locality numbering (hipporag_amd/graph.py,
SURVEY.md 8f-2 -- the reference numbers entity vertices in Python-set order, HippoRAG.py:1159-1187), the NON-baseline
generator variants of hipporag_amd/synth.py, and the host half of the convergence contract
(HippoRAGEngine.retrieve_converged: what is repeated on which state when the engine raises a flag; the reference's
PRPACK simply iterates to 1e-10, HippoRAG.py:1736-1743)."""
import numpy as np
import pytest

import oracle
from hipporag_amd import synth
from hipporag_amd.graph import build_csr, locality_order, locality_score, relabel_csr


def _ppr(csr, reset, damping=0.5):
    import scipy.sparse as sp
    p = sp.csr_matrix((csr.val.astype(np.float64), csr.col_idx, csr.row_ptr), shape=(csr.num_vertices,) * 2)
    return oracle.ppr_exact(p, reset, damping)


def test_locality_order_is_a_permutation_that_keeps_passage_order_and_the_matrix():
    kg = synth.make_kg(3000, 24000, 5, community=64)
    perm = locality_order(kg.csr, kg.passage_vertex)
    v = kg.csr.num_vertices
    assert np.array_equal(np.sort(perm), np.arange(v))
    n_e = v - kg.n_passages
    # passages last, in passage order; entities first, by the first passage that links them
    assert np.array_equal(perm[kg.passage_vertex], n_e + np.arange(kg.n_passages))
    first = np.full(v, np.iinfo(np.int64).max)
    for pos, pvx in enumerate(kg.passage_vertex):
        cols = kg.csr.col_idx[kg.csr.row_ptr[pvx]: kg.csr.row_ptr[pvx + 1]]
        first[cols] = np.minimum(first[cols], pos)
    ents = np.flatnonzero(perm < n_e)
    by_new = ents[np.argsort(perm[ents])]
    assert np.all(np.diff(first[by_new]) >= 0)           # non-decreasing first-linking passage
    same = first[by_new][1:] == first[by_new][:-1]
    assert np.all(by_new[1:][same] > by_new[:-1][same])  # stable inside a passage's group
    # the relabelled matrix is the same operator: PPR commutes with the renaming
    new = relabel_csr(kg.csr, perm)
    assert new.nnz == kg.csr.nnz and np.all(np.diff(new.row_ptr) >= 0)
    for r in range(0, v, 97):
        assert np.all(np.diff(new.col_idx[new.row_ptr[r]: new.row_ptr[r + 1]]) > 0)     # columns sorted, distinct
    rng = np.random.default_rng(1)
    reset = np.zeros(v)
    reset[rng.choice(v, 7, replace=False)] = rng.random(7)
    reset[kg.passage_vertex] += 0.05 * rng.random(kg.n_passages)
    x_old = _ppr(kg.csr, reset)
    reset_new = np.empty(v)
    reset_new[perm] = reset
    x_new = _ppr(new, reset_new)
    np.testing.assert_allclose(x_new[perm], x_old, rtol=1e-12, atol=0)
    if kg.csr.col_sum is not None:
        np.testing.assert_array_equal(new.col_sum[perm], kg.csr.col_sum)


def test_locality_numbering_recovers_what_hash_order_hides_and_finds_nothing_on_the_baseline_generator():
    kg = synth.make_kg(6000, 48000, 9, community=128)          # NON-baseline: communities of 128 entities
    shuffled = synth.hash_order(kg, 3)                          # the reference's situation: ids carry no locality
    assert locality_score(kg.csr, 128) > 0.6 > 0.1 > locality_score(shuffled.csr, 128)
    re = relabel_csr(shuffled.csr, locality_order(shuffled.csr, shuffled.passage_vertex))
    assert locality_score(re, 128) > 0.5
    base = synth.make_kg(6000, 48000, 9)                        # the benchmark generator: uniformly random edges
    re_b = relabel_csr(base.csr, locality_order(base.csr, base.passage_vertex))
    assert locality_score(re_b, 128) < 0.1


def test_hash_order_is_a_pure_renaming_of_the_index():
    kg = synth.make_kg(2000, 16000, 4, community=64)
    sh = synth.hash_order(kg, 11)
    assert sh.n_passages == kg.n_passages and sh.n_facts == kg.n_facts and sh.csr.nnz == kg.csr.nnz
    # degrees (as a multiset), edge weights and the facts' chunk counts survive the renaming
    assert np.array_equal(np.sort(np.diff(sh.csr.row_ptr)), np.sort(np.diff(kg.csr.row_ptr)))
    np.testing.assert_allclose(np.sort(sh.csr.raw), np.sort(kg.csr.raw))
    assert np.array_equal(np.sort(sh.num_chunks), np.sort(kg.num_chunks))
    assert np.array_equal(sh.num_chunks[sh.subj_vertex] > 0, kg.num_chunks[kg.subj_vertex] > 0)


# ------------------------------------------------------------------ retrieve_converged, host half of the contract
class _ScriptedEngine:
    """HippoRAGEngine with the device calls replaced by a script: retrieve() returns what the test queued and records
    (rows, ppr_iters, ppr_max_iters, opt_flags at call time)."""

    def __new__(cls, script, opt_flags=0):
        import torch
        from hipporag_amd.engine import HippoRAGEngine
        self = object.__new__(HippoRAGEngine)
        self.device, self.opt_flags, self.emb_dtype = torch.device("cpu"), opt_flags, torch.float32
        self.calls, script = [], list(script)

        def retrieve(q, kept_idx, kept_score, kept_count, **kw):
            self.calls.append((q[:, 0].tolist(), kw["ppr_iters"], kw["ppr_max_iters"], self.opt_flags))
            return script.pop(0)(q)

        self.retrieve = retrieve
        self._q = lambda x: x
        self.set_flags = lambda bits, on=True: setattr(self, "opt_flags",
                                                       (self.opt_flags | bits) if on else (self.opt_flags & ~bits))
        return self


def _out(q, flags, resid, used):
    import torch
    from hipporag_amd.engine import RetrieveOutput
    b = q.shape[0]
    ids = q[:, :1].to(torch.int32).repeat(1, 3)                  # the query's tag in every slot: merges are visible
    return RetrieveOutput(ids, torch.full((b, 3), float(used[0])), torch.tensor(flags, dtype=torch.int32),
                          torch.tensor(resid, dtype=torch.float32), torch.tensor(used, dtype=torch.int32))


def test_retrieve_converged_repeats_only_the_flagged_queries_on_the_fp32_state_and_restores_the_flags():
    import torch
    from hipporag_amd._lib import FLAG_NOT_CONVERGED, OPT_NO_F16, OPT_NO_FP8, OPT_XCD_BLOCKED
    q = torch.arange(5, dtype=torch.float32).reshape(5, 1)
    k = torch.zeros((5, 5), dtype=torch.int32)
    script = [lambda qq: _out(qq, [0, FLAG_NOT_CONVERGED, 0, FLAG_NOT_CONVERGED, 0], [1e-7, 4e-4, 1e-7, 1e-5, 1e-7],
                              [20, 29, 20, 29, 20]),
              lambda qq: _out(qq, [0, 0], [1e-6, 2e-7], [41, 41])]
    eng = _ScriptedEngine(script, opt_flags=OPT_XCD_BLOCKED)
    out = eng.retrieve_converged(q, k, k.float(), torch.full((5,), 5, dtype=torch.int32), ppr_iters=20, ppr_tol=3e-6,
                                 ppr_max_iters=400)
    assert len(eng.calls) == 2
    rows, iters, max_iters, flags_then = eng.calls[1]
    assert rows == [1.0, 3.0]                                   # those two queries only
    # the residual 4e-4 has to shrink to 3e-6 at damping 0.5: 8 more sweeps than the 29 that ran, + 4 of margin
    assert iters == max_iters == 29 + 8 + 4
    assert flags_then == OPT_XCD_BLOCKED | OPT_NO_FP8 | OPT_NO_F16   # the fp32 state served the repeat ...
    assert eng.opt_flags == OPT_XCD_BLOCKED                          # ... and the engine's own flags are back
    assert out.iters_used.tolist() == [20, 41, 20, 41, 20] and out.flags.tolist() == [0] * 5
    assert out.doc_idx[:, 0].tolist() == [0, 1, 2, 3, 4]             # every row still answers its own query
    np.testing.assert_allclose(out.residual.numpy(), [1e-7, 1e-6, 1e-7, 2e-7, 1e-7], rtol=1e-6)


def test_retrieve_converged_repeats_a_saturated_batch_and_leaves_a_flag_that_ppr_max_iters_cannot_clear():
    import torch
    from hipporag_amd._lib import FLAG_FP8_SATURATED, FLAG_NOT_CONVERGED, OPT_NO_FP8
    q = torch.arange(3, dtype=torch.float32).reshape(3, 1)
    k = torch.zeros((3, 5), dtype=torch.int32)
    cnt = torch.full((3,), 5, dtype=torch.int32)
    script = [lambda qq: _out(qq, [0, FLAG_FP8_SATURATED, 0], [1e-7] * 3, [20] * 3),
              lambda qq: _out(qq, [0, 0, 0], [1e-7] * 3, [20] * 3)]
    eng = _ScriptedEngine(script)
    out = eng.retrieve_converged(q, k, k.float(), cnt, ppr_tol=3e-6)
    assert [c[0] for c in eng.calls] == [[0.0, 1.0, 2.0]] * 2 and eng.calls[1][3] == OPT_NO_FP8   # the WHOLE batch, wider state
    assert eng.opt_flags == 0 and out.flags.tolist() == [0, 0, 0]
    # a query that 30 sweeps cannot settle with ppr_max_iters = 30: no repeat is possible, the flag stays
    script = [lambda qq: _out(qq, [FLAG_NOT_CONVERGED, 0, 0], [1e-3, 1e-7, 1e-7], [30, 20, 20])]
    eng = _ScriptedEngine(script)
    out = eng.retrieve_converged(q, k, k.float(), cnt, ppr_tol=3e-6, ppr_max_iters=30)
    assert len(eng.calls) == 1 and out.flags.tolist() == [FLAG_NOT_CONVERGED, 0, 0]
    # tolerance 0 = the fixed sweep count: a flag is never acted on
    script = [lambda qq: _out(qq, [FLAG_NOT_CONVERGED, 0, 0], [1e-3, 1e-7, 1e-7], [20] * 3)]
    eng = _ScriptedEngine(script)
    eng.retrieve_converged(q, k, k.float(), cnt, ppr_tol=0.0)
    assert len(eng.calls) == 1


def test_looks_undirected_is_what_the_accelerated_plan_is_gated_on():
    """HRAG_OPT_ACCEL (Chebyshev steps: real spectrum) is only honoured on an undirected graph: the wrapper checks that
    the row sums of the adjacency behind the column-normalised CSR equal its column sums (hipporag_amd.graph.
    looks_undirected) -- true for everything build_csr makes (it symmetrises, like the reference's undirected igraph,
    HippoRAG.py:236), false for a CSR whose weights were made asymmetric, for a row shard and without col_sum."""
    from hipporag_amd import synth
    from hipporag_amd.graph import CSRGraph, looks_undirected
    g = synth.make_kg(3000, 30000, 5).csr
    assert looks_undirected(g)
    v = g.val.copy()
    v[g.row_ptr[7]:g.row_ptr[8]] *= 1.5                       # one row re-weighted: A is no longer symmetric
    assert not looks_undirected(CSRGraph(g.num_vertices, g.row_ptr, g.col_idx, v, g.raw, g.col_sum))
    assert not looks_undirected(CSRGraph(g.num_vertices, g.row_ptr, g.col_idx, g.val, g.raw, None))
    assert not looks_undirected(g.rows(0, 1500))
    # small chunks walk the same rows (the 2e8-entry graphs of configs[4] are tested in chunks of 4M entries)
    assert looks_undirected(g, chunk=1000)
    assert not looks_undirected(CSRGraph(g.num_vertices, g.row_ptr, g.col_idx, v, g.raw, g.col_sum), chunk=1000)
    # balanced sums, asymmetric weights: a directed 3-cycle 0 -> 1 -> 2 -> 0 (row sums == column sums == 1) passes
    # the sum test; the sampled mirror-entry test rejects it (no A_ji for A_ij)
    rp = np.array([0, 1, 2, 3], dtype=np.int32)
    cyc = CSRGraph(3, rp, np.array([2, 0, 1], dtype=np.int32), np.ones(3, dtype=np.float32), np.ones(3, dtype=np.float32),
                   np.ones(3))
    assert not looks_undirected(cyc)
    assert looks_undirected(cyc, samples=0)                    # ... which the sums alone would have let through


# ------------------------------------------------------------------ the mirror's batch pipeline (retriever.iter_batched_retrieve)
class _PipelineEngine:
    """Records the order in which the mirror enqueues its device calls.  Phase A answers fact j = 10 * tag + slot for
    the query tagged `tag`; phase B answers documents (tag, kept fact 0, kept count): every row stays traceable."""
    max_batch, max_topk = 4, 3

    def __init__(self, two_halves=True, flag_for=None):
        import torch
        self.log, self.device, self.flag_for = [], torch.device("cpu"), flag_for or {}
        if two_halves:
            self.retrieve_converged_start = self._start

    def score_facts(self, q, k):
        import torch
        self.log.append(("A", q[:, 0].int().tolist()))
        idx = (q[:, :1].int() * 10 + torch.arange(k, dtype=torch.int32)[None, :]).to(torch.int32)
        return idx, torch.linspace(1.0, 0.5, k).repeat(q.shape[0], 1)

    def _answer(self, q, kept_idx, kept_score, kept_count, **kw):
        import torch
        from hipporag_amd.engine import RetrieveOutput
        self.log.append(("B", q[:, 0].int().tolist()))
        tag = q[:, 0].int()
        ids = torch.stack([tag, kept_idx[:, 0], kept_count], 1).to(torch.int32)
        flags = torch.tensor([self.flag_for.get(int(t), 0) for t in tag], dtype=torch.int32)
        return RetrieveOutput(ids, ids.float(), flags, torch.zeros(len(tag)), torch.full((len(tag),), 20, dtype=torch.int32))

    def retrieve_converged(self, *a, **kw):
        return self._answer(*a, **kw)

    def _start(self, *a, **kw):
        from hipporag_amd.engine import PendingRetrieve, host_copy_async
        out = self._answer(*a, **kw)
        eng = self

        class P(PendingRetrieve):
            def finish(self_p):
                eng.log.append(("wait", out.doc_idx[:, 0].tolist()))
                self_p.flags = out.flags.numpy()
                return out
        return P(self, None, out, host_copy_async(out.flags), 0.5, 20, 0.0, 0, False)


def _run_pipeline(eng, n, consume=None, **over):
    import torch
    from hipporag_amd.retriever import identity_rerank_filter, iter_batched_retrieve
    queries = [f"q{i}" for i in range(n)]
    facts = [("s", "p", str(j)) for j in range(10 * n + 10)]
    q_tensor = lambda qs, kind: torch.tensor([[float(q[1:])] for q in qs])
    kw = dict(linking_top_k=2, damping=0.5, passage_node_weight=0.05, ppr_iters=20, num_to_retrieve=3, n_passages=100)
    kw.update(over)
    got = []
    for lo, rows in iter_batched_retrieve(eng, queries, q_tensor, facts, identity_rerank_filter, **kw):
        if consume:
            consume(lo)
        got.append((lo, rows))
    return got


def test_the_batch_pipeline_keeps_the_device_one_batch_ahead_of_the_host():
    eng = _PipelineEngine()
    seen = []
    got = _run_pipeline(eng, 10, consume=lambda lo: seen.append((lo, len(eng.log))))
    tags = lambda lo, hi: list(range(lo, hi))
    assert eng.log == [("A", tags(0, 4)), ("A", tags(4, 8)), ("B", tags(0, 4)), ("A", tags(8, 10)), ("B", tags(4, 8)),
                       ("wait", tags(0, 4)), ("B", tags(8, 10)), ("wait", tags(4, 8)), ("wait", tags(8, 10))]
    # batch 0 reached the consumer only after batch 1's phase B had been enqueued
    assert seen[0] == (0, 6) and [lo for lo, _ in got] == [0, 4, 8]
    for lo, rows in got:
        for i, (d_idx, d_sc, seeds) in enumerate(rows):
            tag = lo + i
            assert d_idx.tolist() == [tag, 10 * tag, 2]                 # its own query, its own best fact, both facts kept
            assert [f[2] for f in seeds] == [str(10 * tag), str(10 * tag + 1)]


def test_the_batch_pipeline_gives_what_the_one_call_form_gives_and_raises_the_references_asserts():
    from hipporag_amd.retriever import batched_retrieve
    a = _run_pipeline(_PipelineEngine(two_halves=True), 9)
    b = _run_pipeline(_PipelineEngine(two_halves=False), 9)      # an engine that only has retrieve_converged (test doubles)
    assert [lo for lo, _ in a] == [lo for lo, _ in b]
    for (_, ra), (_, rb) in zip(a, b):
        for x, y in zip(ra, rb):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2]
    assert _run_pipeline(_PipelineEngine(), 0) == []
    # flag bit 1 (zero-mass reset vector) of query 5: the reference's assert (:1644), raised when ITS batch is finished
    with pytest.raises(AssertionError, match="No phrases found"):
        _run_pipeline(_PipelineEngine(flag_for={5: 2}), 9)
    # no facts in the index: phase A is skipped, every query goes to phase B with nothing kept
    import torch
    from hipporag_amd.retriever import identity_rerank_filter
    eng = _PipelineEngine()
    rows = batched_retrieve(eng, ["q1", "q2"], lambda qs, kind: torch.tensor([[float(q[1:])] for q in qs]), [],
                            identity_rerank_filter, linking_top_k=2, damping=0.5, passage_node_weight=0.05, ppr_iters=20,
                            num_to_retrieve=3, n_passages=100)
    assert [k for k, _ in eng.log] == ["B", "wait"] and rows[1][0].tolist() == [2, -1, 0] and rows[1][2] == []


def test_batch_materialisation_equals_the_per_query_form():
    """retriever.HippoRAG._build_query_solutions: the one-fancy-index fast path (full lists of valid ids, no chunk
    metadata) builds exactly what _build_retrieval_result + QuerySolution build query by query; a batch with a -1
    (fewer passages than asked for) or with chunk metadata takes the per-query form."""
    from hipporag_amd.retriever import BatchRows, HippoRAG
    rag = HippoRAG()
    rag.passage_texts = [f"text {i}" for i in range(50)]
    rag.passage_node_keys = [f"chunk-{i}" for i in range(50)]
    rng = np.random.default_rng(3)

    def rows_of(idx, sc):
        r = BatchRows((idx[i], sc[i], [("a", "b", str(i))] if i % 2 else []) for i in range(len(idx)))
        r.doc_idx, r.doc_score = idx, sc
        return r

    idx = rng.integers(0, 50, (6, 8)).astype(np.int32)
    sc = np.sort(rng.random((6, 8)).astype(np.float32))[:, ::-1].copy()
    qs = [f"q{i}" for i in range(6)]
    for k, meta, hole in ((5, {}, False), (8, {}, False), (20, {}, False), (5, {"chunk-3": {"src": "x"}}, False), (5, {}, True)):
        rag.chunk_metadata = meta
        ii = idx.copy()
        if hole:
            ii[2, 3:] = -1
        rows = rows_of(ii, sc)
        fast = rag._build_query_solutions(qs, rows, k)
        for q, sol, (d_idx, d_sc, seeds) in zip(qs, fast, rows):
            r = rag._build_retrieval_result(q, d_idx, d_sc, k, seeds)
            assert sol.question == q and sol.docs == r.docs and sol.graph_seeds == r.graph_seeds
            assert np.array_equal(sol.doc_scores, r.scores) and sol.doc_metadata == r.doc_metadata
            assert len({id(m) for m in sol.doc_metadata}) == len(sol.doc_metadata)      # one dict per document, as in :505
