"""The host-side mirror of the reference surface (hipporag_amd/retriever.py): what it hands to the
embedding model, how it derives the sweep count, retrieve_ircot, and the behaviour of an index without
facts.  CPU tests use recording stand-ins; GPU tests run the toy corpus of tests/golden."""

import numpy as np
import pytest

from hipporag_amd.retriever import (HippoRAG, QuerySolution, RetrievalConfig, get_query_instruction,
                                    sweeps_for_damping)


class RecordingEmbedder:
    def __init__(self, dim=16):
        self.calls, self.dim = [], dim

    def batch_encode(self, texts, instruction=None, norm=True):
        texts = [texts] if isinstance(texts, str) else list(texts)
        self.calls.append((tuple(texts), instruction, norm))
        rng = np.random.default_rng(len(texts[0]))
        v = rng.standard_normal((len(texts), self.dim)).astype(np.float32)
        return v / np.linalg.norm(v, axis=1, keepdims=True)


def test_queries_are_encoded_under_the_references_instruction_sentences():
    """HippoRAG.py:1414-1423 passes get_query_instruction(...) (prompts/linking.py:1-10), not the lookup
    key: an instruction-tuned embedder must see the same sentences the reference used."""
    emb = RecordingEmbedder()
    rag = HippoRAG(embedding_model=emb)
    rag.get_query_embeddings(["who founded the company?", QuerySolution(question="where?", docs=[])])
    assert [c[1] for c in emb.calls] == [
        "Given a question, retrieve relevant triplet facts that matches this question.",
        "Given a question, retrieve relevant documents that best answer the question."]
    assert all(c[0] == ("who founded the company?", "where?") and c[2] is True for c in emb.calls)
    rag.get_query_embeddings(["where?"])                   # cached: no further encoder call
    assert len(emb.calls) == 2
    assert get_query_instruction("query_to_fact").startswith("Given a question, retrieve relevant triplet facts")
    assert get_query_instruction("anything else") == get_query_instruction("query_to_passage")


def test_sweep_count_follows_the_damping_factor():
    """A fixed 20 sweeps are only enough at the reference's damping 0.5 (0.5^20 ~ 1e-6); the count is
    derived from damping unless the caller pins it."""
    assert sweeps_for_damping(0.5) == 20
    assert sweeps_for_damping(0.85) == 86 and 0.85 ** 86 <= 1e-6 < 0.85 ** 85
    assert sweeps_for_damping(0.3) == 16                   # never below the 16 the reduced-precision states need
    assert HippoRAG(RetrievalConfig(damping=0.85))._ppr_iters() == 86
    assert HippoRAG(RetrievalConfig(damping=0.85, ppr_iters=40))._ppr_iters() == 40
    assert HippoRAG(RetrievalConfig())._ppr_iters() == 20


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_index_without_facts_returns_dpr_results(gpu_device):
    """No triples extracted -> no fact store: the reference returns the dense passage ranking
    (HippoRAG.py:1453-1455, :467-469).  Both through the mirror class and through hrag_retrieve on an
    engine created without fact_desc (every query takes the fallback, flags bit 0)."""
    import torch
    from tests.golden.make_golden import DOCS, QUERIES, MockEmbeddingModel
    from hipporag_amd.engine import HippoRAGEngine
    rag = HippoRAG(RetrievalConfig(embedding_precision="bf16", max_batch=4), embedding_model=MockEmbeddingModel())
    rag.index_from_openie(DOCS, [[] for _ in DOCS])
    sols = rag.retrieve(QUERIES, num_to_retrieve=4)
    dpr = rag.retrieve_dpr(QUERIES, num_to_retrieve=4)
    for s, d in zip(sols, dpr):
        assert s.docs == d.docs and len(s.docs) == 4
        np.testing.assert_allclose(s.doc_scores, d.doc_scores, rtol=0, atol=1e-6)
        assert s.doc_scores[0] == 1.0
    # the C entry point itself, wide batch (fp8-capable engine) and narrow batch
    a = rag._arrays
    for b in (3, 70):
        with HippoRAGEngine(a["csr"], a["passage_vertex"], a["passage_emb"], max_batch=b, max_topk=4) as eng:
            q = torch.randn(b, a["passage_emb"].shape[1], device=gpu_device).to(torch.bfloat16)
            none_i = torch.zeros((b, 5), dtype=torch.int32, device=gpu_device)
            none_s = torch.zeros((b, 5), dtype=torch.float32, device=gpu_device)
            cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)   # ignored: there are no facts
            out = eng.retrieve(q, none_i, none_s, cnt, k=4)
            d_idx, d_sc = eng.dense_retrieve(q, k=4)
            assert torch.all(out.flags == 1)
            assert torch.equal(out.doc_idx, d_idx) and torch.allclose(out.doc_score, d_sc, atol=1e-6, rtol=0)


@pytest.mark.gpu
def test_gpu_retrieve_ircot_matches_the_per_query_loop(gpu_device):
    """retrieve_ircot (HippoRAG.py:509-558) steps all queries in lockstep through batched retrievals; the
    merged rankings must equal the reference's per-query loop (retrieve([query]) / retrieve([thought]) one
    at a time, max-merge of the scores), including the early stop on 'So the answer is:'."""
    from tests.golden.make_golden import DOCS, QUERIES, TRIPLES, MockEmbeddingModel
    rag = HippoRAG(RetrievalConfig(embedding_precision="bf16", max_batch=4, ppr_iters=40), embedding_model=MockEmbeddingModel())
    rag.index_from_openie(DOCS, TRIPLES)
    seen = []

    def reason(query, docs, thoughts):
        seen.append((query, len(docs), tuple(thoughts)))
        if QUERIES.index(query) == 1 and len(thoughts) == 1:
            return "So the answer is: done"
        return f"thought {len(thoughts)} about {docs[0][:24]} for {query}"

    got = rag.retrieve_ircot(QUERIES, max_qa_steps=3, num_to_retrieve=4, reason_fn=reason)
    # the reference's control flow, literally (:524-549)
    for qi, query in enumerate(QUERIES):
        step = rag.retrieve([query], num_to_retrieve=4)[0]
        merged = dict(zip(step.docs, step.doc_scores.tolist()))
        thoughts = []
        for _ in range(1, 3):
            ranked = sorted(merged, key=merged.get, reverse=True)
            thought = reason(query, ranked[:4], thoughts)
            thoughts.append(thought)
            if "So the answer is:" in thought:
                break
            step = rag.retrieve([thought], num_to_retrieve=4)[0]
            for doc, score in zip(step.docs, step.doc_scores.tolist()):
                merged[doc] = max(merged.get(doc, float("-inf")), score)
        items = sorted(merged.items(), key=lambda kv: kv[1], reverse=True)
        assert got[qi].question == query and got[qi].thoughts == thoughts
        assert got[qi].docs == [d for d, _ in items]
        np.testing.assert_allclose(got[qi].doc_scores, [s for _, s in items], rtol=1e-6)
    assert len(got[1].thoughts) == 2 and len(got[0].thoughts) == 2
    with pytest.raises(ValueError):
        rag.retrieve_ircot(QUERIES, max_qa_steps=0)
    assert [s.docs for s in rag.retrieve_ircot(QUERIES, max_qa_steps=1, num_to_retrieve=4)] == \
        [s.docs for s in rag.retrieve(QUERIES, num_to_retrieve=4)]


@pytest.mark.gpu
def test_gpu_num_to_retrieve_beyond_max_topk_returns_the_full_ranking(gpu_device):
    """HippoRAG.py:501-507 slices any prefix of the full ranking: asking for more documents than the engine's device
    top-k holds (max_topk <= 2048) returns them anyway -- the scores of all passages come back (hrag_last_doc_scores)
    and are ranked with the library's rule; the first max_topk entries are the device's own."""
    from tests.golden.make_golden import DOCS, QUERIES, TRIPLES, MockEmbeddingModel
    rag = HippoRAG(RetrievalConfig(embedding_precision="bf16", max_batch=4, retrieval_top_k=3), embedding_model=MockEmbeddingModel())
    rag.index_from_openie(DOCS, TRIPLES)
    few = rag.retrieve(QUERIES, num_to_retrieve=3)
    many = rag.retrieve(QUERIES, num_to_retrieve=7)
    for a, b in zip(few, many):
        assert len(b.docs) == 7 and b.docs[:3] == a.docs
        np.testing.assert_array_equal(np.asarray(b.doc_scores[:3]), np.asarray(a.doc_scores))
        assert np.all(np.diff(np.asarray(b.doc_scores)) <= 0)
    # retrieve_dpr likewise (:704-714): the full dense ranking, the device's own top-k as its prefix
    few_d, many_d = rag.retrieve_dpr(QUERIES, num_to_retrieve=3), rag.retrieve_dpr(QUERIES, num_to_retrieve=7)
    for q, a, b in zip(QUERIES, few_d, many_d):
        assert len(b.docs) == 7 and b.docs[:3] == a.docs
        np.testing.assert_array_equal(np.asarray(b.doc_scores[:3]), np.asarray(a.doc_scores))
        ids, sc = rag.dense_passage_retrieval(q)                       # the per-method seam: the same ranking
        assert b.docs == [rag.passage_texts[i] for i in ids[:7]]
        np.testing.assert_allclose(np.asarray(b.doc_scores), sc[:7], rtol=1e-6)   # B = 1 GEMV vs batch GEMM: other summation order


@pytest.mark.gpu
def test_gpu_a_call_on_another_stream_is_ordered_behind_the_one_in_flight(gpu_device):
    """One call in flight per engine (the workspace belongs to the engine, include/hrag.h): a call on a second stream
    while the previous one has not finished on the device is ordered behind it ON THE DEVICE (hipStreamWaitEvent on the
    previous call's end event) -- no HRAG_EBUSY for a caller with a multi-stream pipeline, no race on the workspace:
    both calls return what a serial run returns.  A second THREAD inside a call is still rejected (host-side flag)."""
    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine
    kg = synth.make_kg(3000, 30000, 5)
    pb, fb = synth.make_embeddings_np(kg.n_passages, 64, 1), synth.make_embeddings_np(kg.n_facts, 64, 2)

    def bf16(bits):
        return torch.from_numpy(bits.view(np.int16)).to(gpu_device).view(torch.bfloat16)

    qa, qb = bf16(synth.make_queries_np(fb, 8, seed=1)[0]), bf16(synth.make_queries_np(fb, 8, seed=2)[0])
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pb, fb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=8, max_topk=10) as eng:
        want_a, want_b = eng.score_facts(qa, k=5), eng.score_facts(qb, k=5)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s1):
            torch.cuda._sleep(400_000_000)              # keeps stream 1 busy for a good while
            got_a = eng.score_facts(qa, k=5)
        with torch.cuda.stream(s2):
            got_b = eng.score_facts(qb, k=5)            # enqueued at once; runs after stream 1's call on the device
            assert not s1.query()                       # ... which had not finished when this call returned
        torch.cuda.synchronize()
        for got, want in ((got_a, want_a), (got_b, want_b)):
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert int(got_b[0].min()) >= 0


@pytest.mark.gpu
def test_gpu_pipelined_batches_equal_the_same_batches_one_call_at_a_time(gpu_device):
    """retrieve() pipelines its batches (phase A of batch i + 1 is enqueued before phase B of batch i, results are waited
    for through events, retriever.iter_batched_retrieve): five batches of a 37-query call -- the last one ragged -- must
    give bit for bit what the same five batches give as five separate calls (nothing of batch i + 1's phase A may
    leak into batch i's phase B through the engine's workspace)."""
    from hipporag_amd import synth
    from hipporag_amd.graph import bf16_bits_to_float
    from tests.helpers import make_case

    class Table:
        def __init__(self, t):
            self.t = t

        def batch_encode(self, texts, instruction=None, norm=True):
            return np.stack([self.t["f" if instruction and "fact" in instruction else "p"][x] for x in texts])

    kg, pass_bits, fact_bits, _ = make_case(6000, 60000, 64, seed=77)
    n = 37
    qf, _ = synth.make_queries_np(fact_bits, n, seed=3)
    qp, _ = synth.make_queries_np(pass_bits, n, seed=4)
    queries = [f"question {i}" for i in range(n)]
    table = {"f": dict(zip(queries, bf16_bits_to_float(qf))), "p": dict(zip(queries, bf16_bits_to_float(qp)))}
    rag = HippoRAG.from_arrays(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                               global_config=RetrievalConfig(embedding_precision="bf16", max_batch=8),
                               embedding_model=Table(table))
    piped = rag.retrieve(queries, num_to_retrieve=20)
    single = []
    for lo in range(0, n, 8):
        single.extend(rag.retrieve(queries[lo: lo + 8], num_to_retrieve=20))
    assert len(piped) == n
    for a, b in zip(piped, single):
        assert a.question == b.question and a.docs == b.docs and a.graph_seeds == b.graph_seeds
        assert np.array_equal(a.doc_scores, b.doc_scores)
    assert len({tuple(s.docs) for s in piped}) > n // 2          # the queries really have different answers
