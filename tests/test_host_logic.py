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
