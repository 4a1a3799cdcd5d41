"""Reference-style CPU retrieval loop, for bench.py's ``cpu_baseline`` leg ("port").

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py).  This mirrors the COST
structure of ``HippoRAG.retrieve()`` (reference src/hipporag/HippoRAG.py:459-480) as closely as
the missing dependencies allow, not just its results:

  * per-query Python loop, one query at a time (:459);
  * fp32 ``np.dot`` against the full fact and passage matrices (:1459, :1496), ``min_max_normalize``,
    full ``np.argsort`` (:1500, :1688, :1746);
  * the seed construction with string node keys and dict lookups, including the loop over ALL
    vertex names in ``get_top_k_weights`` (:1535-1539) and the loop over ALL passages with two
    dict lookups each (:1629-1635);
  * PPR by the single-threaded C port of PRPACK's Gauss-Seidel solver (oracle/prpack_port.c),
    tolerance 1e-10, like igraph's ``implementation='prpack'`` (:1736-1743) -- igraph itself is not
    installable here;
  * the Python-list gather of passage scores (:1745) and the final argsort (:1746).
The LLM filter is the identity (all link_top_k candidates kept), as in the GPU measurement.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
import time
from typing import Dict, List, Tuple

import numpy as np

from .hipporag_ref import RefIndex, min_max_normalize
from .prpack_port import PrpackCSR


class ReferenceStyleRetriever:
    def __init__(self, index: RefIndex):
        self.index = index
        v = index.num_vertices
        n_p = len(index.passage_vertex)
        is_passage = np.zeros(v, dtype=bool)
        is_passage[index.passage_vertex] = True
        # node keys as the reference names them: "entity-<md5>" / "chunk-<md5>" strings
        self.vertex_names: List[str] = [("chunk-%032x" % i) if is_passage[i] else ("entity-%032x" % i)
                                        for i in range(v)]
        self.node_name_to_vertex_idx: Dict[str, int] = {n: i for i, n in enumerate(self.vertex_names)}
        self.passage_node_keys: List[str] = [self.vertex_names[int(i)] for i in index.passage_vertex]
        self.passage_node_idxs: List[int] = [int(i) for i in index.passage_vertex]
        self.ent_node_to_num_chunks: Dict[str, int] = {
            self.vertex_names[i]: int(c) for i, c in enumerate(index.num_chunks) if c > 0}
        self.passage_rows = {k: {"content": k} for k in self.passage_node_keys}
        self.prpack = PrpackCSR(index.p)
        self.ppr_time = 0.0
        self.sim_time = 0.0

    def retrieve_one(self, q_fact: np.ndarray, q_pass: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        ix = self.index
        t0 = time.perf_counter()
        query_fact_scores = min_max_normalize(np.dot(ix.fact_emb, q_fact.T))          # :1459-1461
        link_top_k = ix.linking_top_k
        if len(query_fact_scores) <= link_top_k:                                       # :1683-1688
            candidate_fact_indices = np.argsort(query_fact_scores)[::-1].tolist()
        else:
            candidate_fact_indices = np.argsort(query_fact_scores)[-link_top_k:][::-1].tolist()
        top_k_fact_indices = candidate_fact_indices                                    # identity filter
        self.sim_time += time.perf_counter() - t0

        v = len(self.vertex_names)
        linking_score_map: Dict[str, float] = {}
        phrase_scores: Dict[str, list] = {}
        phrase_weights = np.zeros(v)                                                   # :1577-1579
        passage_weights = np.zeros(v)
        number_of_occurs = np.zeros(v)
        phrases_and_ids = {}   # the reference uses a set (:1581): hash order => ties at the top-k cut are
        #                        undefined upstream; a dict keeps first-occurrence order = the oracle's tie rule
        for rank, f in enumerate(top_k_fact_indices):                                  # :1583-1606
            fact_score = query_fact_scores[f]
            for vid in (int(ix.subj_vertex[f]), int(ix.obj_vertex[f])):
                if vid < 0:
                    continue
                phrase_key = self.vertex_names[vid]
                phrase_id = self.node_name_to_vertex_idx.get(phrase_key, None)
                if phrase_id is not None:
                    weighted_fact_score = fact_score
                    if self.ent_node_to_num_chunks.get(phrase_key, 0) > 0:
                        weighted_fact_score /= self.ent_node_to_num_chunks[phrase_key]
                    phrase_weights[phrase_id] += weighted_fact_score
                    number_of_occurs[phrase_id] += 1
                    phrases_and_ids[(phrase_key, phrase_id)] = None
        phrase_weights = np.divide(phrase_weights, number_of_occurs, out=np.zeros_like(phrase_weights),
                                   where=number_of_occurs != 0)                        # :1608
        for phrase, phrase_id in phrases_and_ids:                                      # :1610-1618
            phrase_scores.setdefault(phrase, []).append(phrase_weights[phrase_id])
        for phrase, scores in phrase_scores.items():
            linking_score_map[phrase] = float(np.mean(scores))
        # get_top_k_weights (:1528-1541): loop over every vertex name
        linking_score_map = dict(sorted(linking_score_map.items(), key=lambda x: x[1], reverse=True)[:link_top_k])
        top_k_phrases_keys = set(linking_score_map.keys())
        for phrase_key in self.node_name_to_vertex_idx:
            if phrase_key not in top_k_phrases_keys:
                phrase_id = self.node_name_to_vertex_idx.get(phrase_key, None)
                if phrase_id is not None:
                    phrase_weights[phrase_id] = 0.0

        t0 = time.perf_counter()
        query_doc_scores = min_max_normalize(np.dot(ix.passage_emb, q_pass.T))        # :1496-1498
        dpr_sorted_doc_ids = np.argsort(query_doc_scores)[::-1]                        # :1500
        dpr_sorted_doc_scores = query_doc_scores[dpr_sorted_doc_ids.tolist()]
        self.sim_time += time.perf_counter() - t0
        normalized = min_max_normalize(dpr_sorted_doc_scores)                          # :1627
        for i, doc_id in enumerate(dpr_sorted_doc_ids.tolist()):                       # :1629-1635
            passage_node_key = self.passage_node_keys[doc_id]
            passage_dpr_score = normalized[i]
            passage_node_id = self.node_name_to_vertex_idx[passage_node_key]
            passage_weights[passage_node_id] = passage_dpr_score * ix.passage_node_weight
            passage_node_text = self.passage_rows[passage_node_key]["content"]
            linking_score_map[passage_node_text] = passage_dpr_score * ix.passage_node_weight
        node_weights = phrase_weights + passage_weights                                # :1638
        if len(linking_score_map) > 30:                                                # :1641-1642
            linking_score_map = dict(sorted(linking_score_map.items(), key=lambda x: x[1], reverse=True)[:30])

        t0 = time.perf_counter()
        reset_prob = np.where(np.isnan(node_weights) | (node_weights < 0), 0, node_weights)   # :1735
        pagerank_scores = self.prpack.solve(reset_prob, ix.damping, "prpack")[0].tolist()     # :1736
        doc_scores = np.array([pagerank_scores[idx] for idx in self.passage_node_idxs])       # :1745
        sorted_doc_ids = np.argsort(doc_scores)[::-1]                                          # :1746
        sorted_doc_scores = doc_scores[sorted_doc_ids.tolist()]
        self.ppr_time += time.perf_counter() - t0
        return sorted_doc_ids, sorted_doc_scores

    def retrieve(self, q_fact: np.ndarray, q_pass: np.ndarray):
        return [self.retrieve_one(q_fact[i], q_pass[i]) for i in range(q_fact.shape[0])]


# --------------------------------------------------------------------------------------------
# "vectorised" leg: what a tuned CPU deployment of the same algorithm would do
# --------------------------------------------------------------------------------------------
_SPMM_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_spmm.c")
_SPMM_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhrag_cpu_spmm.so")
_spmm = None


def build_spmm(force: bool = False) -> str:
    """gcc -O3 -fopenmp -shared -fPIC oracle/cpu_spmm.c -> oracle/libhrag_cpu_spmm.so"""
    if force or not os.path.exists(_SPMM_LIB) or os.path.getmtime(_SPMM_LIB) < os.path.getmtime(_SPMM_SRC):
        subprocess.check_call(["gcc", "-O3", "-std=c11", "-fopenmp", "-shared", "-fPIC", "-o", _SPMM_LIB, _SPMM_SRC])
    return _SPMM_LIB


def _load_spmm():
    global _spmm
    if _spmm is None:
        build_spmm()
        lib = ctypes.CDLL(_SPMM_LIB)
        f32p, i64p, i32p = (ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64),
                            ctypes.POINTER(ctypes.c_int32))
        lib.hro_spmm_f32.argtypes = [ctypes.c_int64, i64p, i32p, f32p, f32p, f32p, ctypes.c_float,
                                     ctypes.c_float, ctypes.c_int32, f32p]
        lib.hro_spmm_f32.restype = None
        _spmm = lib
    return _spmm


class VectorisedRetriever:
    """Batched CPU path: one fp32 sgemm per similarity stage (all BLAS threads), argpartition instead of
    full argsorts, numpy seed arithmetic, and the fixed-sweep power iteration as an OpenMP SpMM over the
    whole batch (oracle/cpu_spmm.c).  Same algorithm and sweep count as the GPU path; results agree with
    the exact solution to ~1e-6 (20 sweeps at damping 0.5)."""

    MAX_B = 256

    def __init__(self, index: RefIndex):
        self.index = index
        p = index.p.tocsr()
        self.n = p.shape[0]
        self.rowptr = np.ascontiguousarray(p.indptr, dtype=np.int64)
        self.col = np.ascontiguousarray(p.indices, dtype=np.int32)
        self.val = np.ascontiguousarray(p.data, dtype=np.float32)
        self.sim_time = self.seed_time = self.ppr_time = self.rank_time = 0.0

    def _spmm(self, x, v, alpha, y):
        f32p = ctypes.POINTER(ctypes.c_float)
        _load_spmm().hro_spmm_f32(self.n, self.rowptr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                  self.col.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                  self.val.ctypes.data_as(f32p), x.ctypes.data_as(f32p), v.ctypes.data_as(f32p),
                                  alpha, 1.0 - alpha, x.shape[1], y.ctypes.data_as(f32p))

    def retrieve(self, q_fact: np.ndarray, q_pass: np.ndarray, iters: int = 20, k: int = 200):
        ix = self.index
        b = q_fact.shape[0]
        assert b <= self.MAX_B
        t0 = time.perf_counter()
        sf = q_fact.astype(np.float32) @ ix.fact_emb.T                      # [B, F] one sgemm (:1459)
        sp_ = q_pass.astype(np.float32) @ ix.passage_emb.T                  # [B, Np]      (:1496)
        self.sim_time += time.perf_counter() - t0
        t0 = time.perf_counter()
        kf = ix.linking_top_k
        v = np.zeros((self.n, b), dtype=np.float32)
        mn, mx = sp_.min(1, keepdims=True), sp_.max(1, keepdims=True)
        rng = np.where(mx - mn == 0, 1, mx - mn)
        v[ix.passage_vertex, :] = ((sp_ - mn) / rng).T * np.float32(ix.passage_node_weight)   # :1626-1635
        part = np.argpartition(sf, -kf, axis=1)[:, -kf:]                    # :1683-1688 without the full sort
        for q in range(b):
            row = sf[q]
            cand = part[q][np.argsort(row[part[q]], kind="stable")[::-1]]
            lo, hi = row.min(), row.max()
            qfs = {int(f): np.float32((row[f] - lo) / (hi - lo)) if hi > lo else np.float32(1) for f in cand}
            w, cnt, order = {}, {}, []
            for f in cand:                                                   # :1583-1606
                for vid in (int(ix.subj_vertex[f]), int(ix.obj_vertex[f])):
                    if vid < 0:
                        continue
                    s = qfs[int(f)]
                    nc = int(ix.num_chunks[vid])
                    if nc > 0:
                        s = s / nc
                    w[vid] = w.get(vid, 0.0) + float(s)
                    cnt[vid] = cnt.get(vid, 0) + 1
                    if vid not in order:
                        order.append(vid)
            items = sorted(((vid, w[vid] / cnt[vid]) for vid in order), key=lambda t: t[1], reverse=True)[:kf]
            for vid, wt in items:                                            # :1638
                v[vid, q] += np.float32(wt)
        self.seed_time += time.perf_counter() - t0
        t0 = time.perf_counter()
        v /= v.sum(0, keepdims=True)
        x, y = v.copy(), np.empty_like(v)
        for _ in range(iters):                                               # :1736-1743, fixed sweep count
            self._spmm(x, v, np.float32(ix.damping), y)
            x, y = y, x
        self.ppr_time += time.perf_counter() - t0
        t0 = time.perf_counter()
        doc = (x[ix.passage_vertex, :] / x.sum(0, keepdims=True, dtype=np.float64)).T   # [B, Np]  (:1745)
        kk = min(k, doc.shape[1])
        top = np.argpartition(doc, -kk, axis=1)[:, -kk:]
        ids = np.empty((b, kk), dtype=np.int64)
        for q in range(b):                                                   # :1746 on the k survivors
            ids[q] = top[q][np.lexsort((top[q], doc[q, top[q]]))[::-1]]
        scores = np.take_along_axis(doc, ids, 1)
        self.rank_time += time.perf_counter() - t0
        return ids, scores


def networkx_pagerank_leg(index: RefIndex, resets: np.ndarray, budget_s: float):
    """PPR through networkx.pagerank(tol=1e-10) (networkx 3.4.2 is what the reference's requirements list;
    igraph itself is not installable here).  resets: [n_queries, V].  Returns seconds per query and the
    max relative difference to the PRPACK port on the passage vertices."""
    import networkx as nx
    t0 = time.perf_counter()
    p = index.p.tocoo()
    g = nx.DiGraph()
    g.add_nodes_from(range(index.num_vertices))
    # column j of P = out-edges of j (P is column-stochastic): edge j -> i with weight P[i, j]
    g.add_weighted_edges_from(zip(p.col.tolist(), p.row.tolist(), p.data.tolist()))
    build_s = time.perf_counter() - t0
    prp = PrpackCSR(index.p)
    n_done, worst, t1 = 0, 0.0, time.perf_counter()
    for r in resets:
        pers = {int(i): float(r[i]) for i in np.flatnonzero(r)}
        pr = nx.pagerank(g, alpha=index.damping, personalization=pers, tol=1e-10, max_iter=1000, weight="weight")
        x = np.fromiter((pr[i] for i in index.passage_vertex.tolist()), dtype=np.float64)
        ref = prp.solve(r, index.damping, "prpack")[0][index.passage_vertex]
        worst = max(worst, float(np.abs(x / ref - 1).max()))
        n_done += 1
        if time.perf_counter() - t1 > budget_s:
            break
    return {"s_per_query": (time.perf_counter() - t1) / max(n_done, 1), "queries": n_done,
            "graph_build_s": build_s, "max_rel_diff_vs_prpack_port": worst}
