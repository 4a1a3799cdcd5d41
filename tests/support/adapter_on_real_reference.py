"""Run by tests/test_adapter_real_reference.py in a subprocess: attach() / detach() on a REAL reference
HippoRAG object (tests/golden/ref_harness.py: the reference package imported from /root/reference/src,
igraph / LLM / embedding model substituted), with the device engine replaced by a CPU stand-in built on
the oracle -- what is under test is the adapter's contact surface with the real class (every attribute it
reads, the stores' row format, the config fields, the result classes, the restore on detach), not kernels."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_golden as mg          # noqa: E402
import ref_harness as rh          # noqa: E402
import oracle                      # noqa: E402
from hipporag_amd import engine as engine_mod, reference_adapter as ra      # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float, float_to_bf16_bits       # noqa: E402


class Bf16Mock(mg.MockEmbeddingModel):
    """bf16-representable fp32 vectors: the reference (fp32 numpy) and the adapter (bf16 at the device
    boundary) then see identical inputs."""

    def batch_encode(self, texts, instruction=None, norm=True):
        e = super().batch_encode(texts, instruction=instruction, norm=norm)
        return bf16_bits_to_float(float_to_bf16_bits(e)).astype(np.float32)


class OracleEngine:
    """The HippoRAGEngine surface the adapter uses, computed by the oracle on the CPU."""

    def __init__(self, csr, passage_vertex, passage_emb, fact_emb=None, subj_vertex=None, obj_vertex=None,
                 num_chunks=None, *, max_batch=256, max_topk=200, **_):
        import scipy.sparse as sp
        p = sp.csr_matrix((csr.val.astype(np.float64), csr.col_idx, csr.row_ptr), shape=(csr.num_vertices,) * 2)
        as_f32 = lambda e: np.asarray(e, np.float32) if e.dtype == np.float32 else bf16_bits_to_float(e)   # fp32-faithful / bf16 bits
        self.index = oracle.RefIndex(as_f32(fact_emb), as_f32(passage_emb), subj_vertex,
                                     obj_vertex, num_chunks, np.asarray(passage_vertex), p)
        self.device, self.max_batch, self.max_topk = torch.device("cpu"), max_batch, max_topk
        self.closed = False
        # like HippoRAGEngine: fp32 matrices -> the queries stay fp32 (HRAG_F32_SPLIT), bf16 bits -> bf16 queries
        self.emb_dtype = torch.float32 if np.asarray(passage_emb).dtype == np.float32 else torch.bfloat16

    def score_facts(self, q, k=5):
        q = q.float().numpy()
        idx = np.full((len(q), k), -1, np.int32)
        sc = np.zeros((len(q), k), np.float32)
        for i, row in enumerate(q):
            s = oracle.fact_scores(self.index.fact_emb, row)
            top = oracle.topk_desc(s, k)
            idx[i, :len(top)], sc[i, :len(top)] = top, s[top]
        return torch.from_numpy(idx), torch.from_numpy(sc)

    opt_flags = 0

    def retrieve(self, q_pass, kept_idx, kept_score, kept_count, *, link_top_k, damping, passage_node_weight,
                 ppr_iters, k, ppr_tol=0.0, ppr_max_iters=0):
        import dataclasses
        ix = dataclasses.replace(self.index, linking_top_k=link_top_k, damping=damping,
                                 passage_node_weight=passage_node_weight)
        b = q_pass.shape[0]
        d_idx = np.full((b, k), -1, np.int32)
        d_sc = np.zeros((b, k), np.float32)
        flags = np.zeros(b, np.int32)
        for i in range(b):
            n = int(kept_count[i])
            dpr_ids, dpr_sc = oracle.dense_passage_scores(ix.passage_emb, q_pass[i].float().numpy())
            if n == 0:
                flags[i] = 1
                ids, sc = dpr_ids, dpr_sc
            else:
                scores = np.zeros(len(ix.subj_vertex), np.float32)
                kept = kept_idx[i, :n].numpy()
                scores[kept] = kept_score[i, :n].numpy()
                sid, sw = oracle.seed_weights(ix, scores, kept.tolist(), link_top_k)
                by_p = np.empty_like(dpr_sc)
                by_p[dpr_ids] = dpr_sc
                # with a tolerance the device iterates until the passage scores stand still: the exact solve stands in
                ids, sc, _ = oracle.run_ppr(ix, oracle.reset_vector(ix, sid, sw, by_p), damping,
                                            "exact" if ppr_tol > 0 else "power", ppr_iters)
            d_idx[i, :min(k, len(ids))], d_sc[i, :min(k, len(ids))] = ids[:k], sc[:k]
        return engine_mod.RetrieveOutput(torch.from_numpy(d_idx), torch.from_numpy(d_sc), torch.from_numpy(flags))

    retrieve_converged = retrieve

    def dense_retrieve(self, q_pass, k=200):            # the mirror's retrieve_dpr (not used by the adapter)
        b = q_pass.shape[0]
        d_idx, d_sc = np.full((b, k), -1, np.int32), np.zeros((b, k), np.float32)
        for i in range(b):
            ids, sc = oracle.dense_passage_scores(self.index.passage_emb, q_pass[i].float().numpy())
            d_idx[i, :min(k, len(ids))], d_sc[i, :min(k, len(ids))] = ids[:k], sc[:k]
        return torch.from_numpy(d_idx), torch.from_numpy(d_sc)

    f32_split = False

    @property
    def dim(self):
        return self.index.passage_emb.shape[1]

    def sim_scores(self, which, q):
        emb = self.index.fact_emb if which == "facts" else self.index.passage_emb
        return torch.from_numpy((q.float().numpy().astype(np.float64) @ emb.T.astype(np.float64)).astype(np.float32))

    def ppr(self, reset, damping=0.5, iters=20):
        x = np.stack([oracle.ppr_power(self.index.p, r.numpy().astype(np.float64), damping, iters) for r in reset])
        return torch.from_numpy(x.astype(np.float32)), torch.zeros(len(reset), dtype=torch.int32)

    def set_flags(self, *a):
        pass

    def close(self):
        self.closed = True


def main():
    engine_mod.HippoRAGEngine = OracleEngine          # the adapter imports it from hipporag_amd.engine at call time
    tmp = tempfile.mkdtemp(prefix="hrag_adapter_")
    rag = rh.build_reference_rag(tmp, mg.DOCS, mg.TRIPLES, Bf16Mock())
    cls = type(rag)
    before = rag.retrieve(list(mg.QUERIES), num_to_retrieve=5)           # the reference's own path
    ra.attach(rag, max_batch=2)                                           # 3 queries -> two device batches
    for name in ("get_fact_scores", "dense_passage_retrieval", "run_ppr", "retrieve"):
        assert name in rag.__dict__, name                                 # instance attributes shadow the class
    after, metrics = rag.retrieve(list(mg.QUERIES), num_to_retrieve=5, gold_docs=[[mg.DOCS[1]], [mg.DOCS[3]], [mg.DOCS[6]]])
    from hipporag.utils.misc_utils import QuerySolution                   # the reference's own result class
    for a, b in zip(after, before):
        assert isinstance(a, QuerySolution) and a.question == b.question
        assert a.docs == b.docs, (a.docs, b.docs)
        np.testing.assert_allclose(a.doc_scores, b.doc_scores, rtol=2e-5)
        assert len(a.graph_seeds) == 5 and isinstance(a.doc_metadata, list)
    assert "Recall@1" in metrics and "Recall@5" in metrics                # the reference's RetrievalRecall ran
    # per-query seams against the reference's own methods
    q = mg.QUERIES[0]
    fs_ref = cls.get_fact_scores(rag, q)
    np.testing.assert_allclose(rag.get_fact_scores(q), fs_ref, atol=2e-6, rtol=0)
    ids_ref, sc_ref = cls.dense_passage_retrieval(rag, q)
    ids, sc = rag.dense_passage_retrieval(q)
    np.testing.assert_allclose(sc, sc_ref, atol=2e-6, rtol=0)
    assert ids[0] == ids_ref[0]
    assert rag.ppr_time > 0 and rag.all_retrieval_time > 0
    eng = rag._mi355x["engine"]
    ra.detach(rag)
    assert eng.closed and rag._mi355x is None
    for name in ("get_fact_scores", "dense_passage_retrieval", "run_ppr", "retrieve"):
        assert name not in rag.__dict__ and getattr(rag, name).__func__ is getattr(cls, name), name
    again = rag.retrieve(list(mg.QUERIES), num_to_retrieve=5)
    assert [s.docs for s in again] == [s.docs for s in before]
    print("ADAPTER_ON_REAL_REFERENCE_OK")


if __name__ == "__main__":
    main()
