"""Reference-side binding: switch an indexed ``hipporag.HippoRAG`` object onto the MI355X engine.

The reference (OSU-NLP-Group/HippoRAG, pure Python) has no plugin / FFI layer on this path; its seams
are methods of ``class HippoRAG`` (reference src/hipporag/HippoRAG.py).  ``attach(rag)`` builds the
device index from the object's own state and replaces exactly those methods -- nothing else of the
reference changes, and ``detach(rag)`` restores them:

    per-query seams (keep the reference's loop at :459, win the kernel time)
        get_fact_scores(query)             :1427-1465   -> hrag_sim_scores + min-max
        dense_passage_retrieval(query)     :1467-1502   -> hrag_sim_scores + min-max + argsort
        run_ppr(reset_prob, damping)       :1709-1749   -> hrag_ppr
    batched route (``batched_retrieve=True``, default)
        retrieve(queries, num_to_retrieve, gold_docs)  :413-499 -> hrag_score_facts / rerank_filter on
                                                                    the host / hrag_retrieve

``rag`` is duck-typed; what is read from it (all set by the reference's own
``prepare_retrieval_objects``, :1287-1389):
    graph (igraph.Graph: vcount(), get_edgelist(), es["weight"]), node_name_to_vertex_idx,
    passage_node_idxs, passage_node_keys, fact_node_keys, passage_embeddings, fact_embeddings,
    ent_node_to_chunk_ids, fact_embedding_store.get_rows(keys), chunk_embedding_store.get_row(key),
    chunk_metadata, global_config.{retrieval_top_k, linking_top_k, damping, passage_node_weight},
    query_to_embedding, get_query_embeddings(queries), rerank_filter(query, facts, indices, len_after_rerank=)
"""

from __future__ import annotations

import ast
import logging
import time
from hashlib import md5
from typing import List, Optional

import numpy as np

from .graph import build_csr, float_to_bf16_bits

logger = logging.getLogger("hipporag_amd")

_PATCHED = ("get_fact_scores", "dense_passage_retrieval", "run_ppr", "retrieve")


def _min_max_normalize(x: np.ndarray) -> np.ndarray:          # utils/misc_utils.py:130-139
    mn, mx = np.min(x), np.max(x)
    rng = mx - mn
    return np.ones_like(x) if rng == 0 else (x - mn) / rng


def _entity_key(phrase: str) -> str:                          # compute_mdhash_id(..., "entity-"), misc_utils.py:141-152
    return "entity-" + md5(phrase.encode()).hexdigest()


def _solution_types():
    """The reference's own result classes when its package imports, ours otherwise (same fields)."""
    try:
        from hipporag.utils.misc_utils import QuerySolution, RetrievalResult  # type: ignore
        return QuerySolution, RetrievalResult
    except Exception:
        from .retriever import QuerySolution, RetrievalResult
        return QuerySolution, RetrievalResult


def index_arrays_from_reference(rag) -> dict:
    """The integer / float arrays the engine needs, read off an indexed reference object
    (prepare_retrieval_objects must have run).  Also what tests/golden/make_ref_golden.py stores."""
    g = rag.graph
    v = int(g.vcount())
    es = np.asarray(g.get_edgelist(), dtype=np.int64).reshape(-1, 2)
    w = np.asarray(g.es["weight"], dtype=np.float64) if es.shape[0] else np.zeros(0)
    key2v = rag.node_name_to_vertex_idx
    fact_keys = list(rag.fact_node_keys)
    rows = rag.fact_embedding_store.get_rows(fact_keys) if fact_keys else {}
    facts = [ast.literal_eval(rows[k]["content"]) for k in fact_keys]     # the reference eval()s (:1693)
    subj = np.array([key2v.get(_entity_key(f[0].lower()), -1) for f in facts], np.int32)   # :1584-1597
    obj = np.array([key2v.get(_entity_key(f[2].lower()), -1) for f in facts], np.int32)
    nchunks = np.zeros(v, np.int32)
    for k, s in (rag.ent_node_to_chunk_ids or {}).items():   # divisor of :1600-1601
        if k in key2v:
            nchunks[key2v[k]] = len(s)
    has_facts = len(facts) > 0
    return {"num_vertices": v, "edge_src": es[:, 0].copy(), "edge_dst": es[:, 1].copy(), "edge_w": w,
            "passage_vertex": np.asarray(rag.passage_node_idxs, np.int32),
            "passage_emb": np.asarray(rag.passage_embeddings, np.float32),
            "fact_emb": np.asarray(rag.fact_embeddings, np.float32) if has_facts else None,
            "subj_vertex": subj, "obj_vertex": obj, "num_chunks": nchunks, "facts": facts}


def build_engine_from_reference(rag, *, max_batch: int = 256, embedding_precision: str = "f32", accel: bool = True):
    """Device index from the reference object's host state (prepare_retrieval_objects must have run).
    embedding_precision "f32" (default): the reference's fp32 matrices as they are (HippoRAG.py:1342-1345) on the
    fp32-faithful engine (HRAG_F32_SPLIT); "bf16": rounded to bf16, a third of the embedding stream."""
    from . import _lib
    from .engine import HippoRAGEngine
    a = index_arrays_from_reference(rag)
    facts = a["facts"]
    csr = build_csr(a["num_vertices"], a["edge_src"], a["edge_dst"], a["edge_w"])   # edge rules of HippoRAG.py:1189-1223
    pv = a["passage_vertex"]
    has_facts = len(facts) > 0
    cfg = rag.global_config
    if embedding_precision not in ("f32", "bf16"):
        raise ValueError("embedding_precision must be 'f32' or 'bf16'")
    conv = (lambda e: np.ascontiguousarray(e, np.float32)) if embedding_precision == "f32" else float_to_bf16_bits
    eng = HippoRAGEngine(csr, pv, conv(a["passage_emb"]),
                         conv(a["fact_emb"]) if has_facts else None,
                         a["subj_vertex"] if has_facts else None, a["obj_vertex"] if has_facts else None,
                         a["num_chunks"] if has_facts else None, max_batch=max_batch, locality="auto",
                         flags=_lib.OPT_ACCEL if accel else 0,   # Chebyshev steps in the fp8 stages (undirected graph: :236)
                         max_topk=int(min(2048, max(1, min(cfg.retrieval_top_k, len(pv))))))
    return eng, facts


def attach(rag, *, max_batch: int = 256, ppr_iters: Optional[int] = None, batched_retrieve: bool = True,
           ppr_tol: float = 1.5e-6, ppr_max_iters: int = 400, embedding_precision: str = "f32",
           ppr_base_iters_narrow: Optional[int] = 12):
    """Patch ``rag`` in place; returns it.  Call again after ``index()`` / ``delete()`` (they change
    the graph and the stores; the reference only resets ``ready_to_retrieve`` on delete, :411)."""
    import torch
    if getattr(rag, "_mi355x", None) is not None:
        detach(rag)
    if not getattr(rag, "ready_to_retrieve", False):
        rag.prepare_retrieval_objects()
    eng, facts = build_engine_from_reference(rag, max_batch=max_batch, embedding_precision=embedding_precision)
    cfg = rag.global_config
    from .retriever import sweeps_for_damping
    # fixed sweep count of the power iteration: given, or derived from the damping factor (20 at 0.5)
    sweeps = int(ppr_iters) if ppr_iters else sweeps_for_damping(float(cfg.damping))
    saved = {name: rag.__dict__.get(name, None) for name in _PATCHED}
    QuerySolution, RetrievalResult = _solution_types()
    dev = eng.device

    def q_tensor(queries: List[str], kind: str):
        m = np.stack([np.asarray(rag.query_to_embedding[kind][q], np.float32).reshape(-1) for q in queries])
        return torch.from_numpy(m).to(dev).to(getattr(eng, "emb_dtype", torch.bfloat16))

    def get_fact_scores(query: str) -> np.ndarray:                        # :1427-1465
        if len(facts) == 0:
            return np.array([])
        rag.get_query_embeddings([query])
        try:
            s = eng.sim_scores("facts", q_tensor([query], "triple"))[0].cpu().numpy()
            return _min_max_normalize(s)
        except Exception as exc:                                          # :1463-1465
            logger.error("Error computing fact scores: %s", exc)
            return np.array([])

    def dense_passage_retrieval(query: str):                              # :1467-1502
        rag.get_query_embeddings([query])
        s = eng.sim_scores("passages", q_tensor([query], "passage"))[0].cpu().numpy()
        s = _min_max_normalize(s)
        ids = np.argsort(s, kind="stable")[::-1]
        return ids, s[ids]

    def run_ppr(reset_prob: np.ndarray, damping: float = 0.5):            # :1709-1749
        t0 = time.time()
        if damping is None:
            damping = 0.5
        r = torch.from_numpy(np.asarray(reset_prob, dtype=np.float32).reshape(1, -1))
        x, flags = eng.ppr(r, damping=damping, iters=int(ppr_iters) if ppr_iters else sweeps_for_damping(float(damping)))
        if int(flags[0].item()) & 2:
            raise ValueError("reset vector has no positive entry")        # igraph raises here
        doc = x[0].cpu().numpy()[np.asarray(rag.passage_node_idxs)]
        ids = np.argsort(doc, kind="stable")[::-1]
        rag.ppr_time = getattr(rag, "ppr_time", 0.0) + time.time() - t0
        return ids, doc[ids]

    def build_result(query, ids, scores, num_to_retrieve, seeds):         # :501-507
        ids = [int(i) for i in ids[:num_to_retrieve] if i >= 0]
        keys = [rag.passage_node_keys[i] for i in ids]
        docs = [rag.chunk_embedding_store.get_row(k)["content"] for k in keys]
        meta = [dict(getattr(rag, "chunk_metadata", {}).get(k, {})) for k in keys]
        return RetrievalResult(query=query, docs=docs, scores=np.asarray(scores[:len(ids)]),
                               doc_metadata=meta, graph_seeds=seeds or [])

    def retrieve(queries: List[str], num_to_retrieve: Optional[int] = None, gold_docs=None):   # :413-499
        from .retriever import gc_paused, iter_batched_retrieve
        t_start = time.time()
        if num_to_retrieve is None:
            num_to_retrieve = cfg.retrieval_top_k
        rag.get_query_embeddings(queries)
        results = []
        # batch by batch: one batch's document lists are built while the device works on the next one
        for lo, rows in iter_batched_retrieve(eng, queries, q_tensor, facts, rag.rerank_filter,
                                              linking_top_k=int(cfg.linking_top_k), damping=cfg.damping,
                                              passage_node_weight=cfg.passage_node_weight, ppr_iters=sweeps,
                                              num_to_retrieve=int(num_to_retrieve), n_passages=len(rag.passage_node_keys),
                                              timers=rag, ppr_tol=ppr_tol, ppr_max_iters=ppr_max_iters,
                                              ppr_base_iters_narrow=ppr_base_iters_narrow):
            with gc_paused():
                for q, (d_idx, d_sc, seeds) in zip(queries[lo: lo + len(rows)], rows):
                    r = build_result(q, d_idx, d_sc, num_to_retrieve, seeds)
                    results.append(QuerySolution(question=r.query, docs=r.docs, doc_scores=r.scores,
                                                 doc_metadata=r.doc_metadata, graph_seeds=r.graph_seeds))
        rag.all_retrieval_time = getattr(rag, "all_retrieval_time", 0.0) + time.time() - t_start
        if gold_docs is not None:
            try:
                from hipporag.evaluation.retrieval_eval import RetrievalRecall  # type: ignore
                overall, _ = RetrievalRecall(global_config=cfg).calculate_metric_scores(
                    gold_docs=gold_docs, retrieved_docs=[r.docs for r in results],
                    k_list=[1, 2, 5, 10, 20, 30, 50, 100, 150, 200])
            except Exception:
                from .retriever import HippoRAG as _Mirror
                overall = _Mirror._recall(gold_docs, [r.docs for r in results])
            return results, overall
        return results

    rag.get_fact_scores = get_fact_scores
    rag.dense_passage_retrieval = dense_passage_retrieval
    rag.run_ppr = run_ppr
    if batched_retrieve:
        rag.retrieve = retrieve
    rag._mi355x = {"engine": eng, "saved": saved, "facts": facts}
    return rag


def detach(rag):
    """Undo ``attach``: restore the reference methods and free the device index."""
    st = getattr(rag, "_mi355x", None)
    if st is None:
        return rag
    for name, old in st["saved"].items():
        if old is None:
            rag.__dict__.pop(name, None)        # fall back to the class attribute
        else:
            rag.__dict__[name] = old
    st["engine"].close()
    rag._mi355x = None
    return rag
