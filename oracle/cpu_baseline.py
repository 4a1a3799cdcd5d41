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
