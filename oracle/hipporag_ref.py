"""Oracle: array-level restatement of ``HippoRAG.retrieve()``'s per-query path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against recordings of the
reference's own code (tests/golden/ref_*.npz, tests/test_ref_golden.py) for every stage
except the PPR arithmetic, which is PARITY UNPINNED (igraph absent; oracle/ppr.py).

The reference works on strings (fact triples, md5 node keys) and an igraph
object; this restatement works on the integer arrays those resolve to:

    fact f        -> (subj_vertex[f], obj_vertex[f])   md5("entity-"+phrase.lower())
                     looked up in node_name_to_vertex_idx, -1 when absent
                     (HippoRAG.py:1584-1597, utils/misc_utils.py:141-152)
    entity vertex -> num_chunks[v] = len(ent_node_to_chunk_ids[key])
                     (HippoRAG.py:1600-1601, built at :867-913)
    passage p     -> passage_vertex[p] = passage_node_idxs[p]   (HippoRAG.py:1333)

Tie rule (undefined upstream: ``np.argsort`` default kind is not stable and
``get_top_k_weights`` sorts a dict filled from a ``set``): the oracle DEFINES
  * rankings:   score descending, then index descending
                == ``np.argsort(x, kind='stable')[::-1]``
  * seed top-k: weight descending, then first occurrence in
                (fact rank, subject-before-object) order.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from . import ppr as _ppr


# --------------------------------------------------------------------------- #
# utils/misc_utils.py:130-139
# --------------------------------------------------------------------------- #
def min_max_normalize(x: np.ndarray) -> np.ndarray:
    """(x - min) / (max - min); all-equal input -> ones (misc_utils.py:130-139).
    dtype is preserved (fp32 in -> fp32 arithmetic, like the reference)."""
    x = np.asarray(x)
    min_val = np.min(x)
    max_val = np.max(x)
    range_val = max_val - min_val
    if range_val == 0:
        return np.ones_like(x)
    return (x - min_val) / range_val


@dataclass
class RefIndex:
    """The arrays ``prepare_retrieval_objects`` (HippoRAG.py:1287-1389) stages."""
    fact_emb: np.ndarray          # fp32 [F, D]   (self.fact_embeddings, :1345)
    passage_emb: np.ndarray       # fp32 [Np, D]  (self.passage_embeddings, :1343)
    subj_vertex: np.ndarray       # int  [F]
    obj_vertex: np.ndarray        # int  [F]
    num_chunks: np.ndarray        # int  [V]      (0 => no division, :1600)
    passage_vertex: np.ndarray    # int  [Np]     (self.passage_node_idxs, :1333)
    p: sp.csr_matrix              # column-normalised adjacency (oracle/ppr.py)
    linking_top_k: int = 5        # config_utils.py:184
    passage_node_weight: float = 0.05   # config_utils.py:91
    damping: float = 0.5          # config_utils.py:192
    retrieval_top_k: int = 200    # config_utils.py:188

    @property
    def num_vertices(self) -> int:
        return self.p.shape[0]


# --------------------------------------------------------------------------- #
# similarity: HippoRAG.py:1427-1465 (facts), :1467-1502 (passages)
# --------------------------------------------------------------------------- #
def _dot(emb: np.ndarray, q: np.ndarray, exact: bool) -> np.ndarray:
    """``np.dot(embeddings, q.T)`` (HippoRAG.py:1459,1496).

    exact=False: literal fp32 BLAS dot (what the reference runs).
    exact=True : fp64 accumulation rounded once to fp32 -- the summation-order
                 independent value both the fp32 BLAS path and the MFMA path
                 approximate; used as the parity target for device scores.
    """
    if exact:
        return (emb.astype(np.float64) @ np.asarray(q, dtype=np.float64).T).astype(np.float32)
    return np.dot(emb, q.T)


def fact_scores(fact_emb: np.ndarray, q: np.ndarray, exact: bool = True) -> np.ndarray:
    """get_fact_scores (HippoRAG.py:1427-1465): dot, squeeze, min-max; no facts -> []."""
    if len(fact_emb) == 0:
        return np.array([])
    s = _dot(fact_emb, q, exact)
    s = np.squeeze(s) if s.ndim == 2 else s
    return min_max_normalize(s)


def topk_desc(x: np.ndarray, k: Optional[int] = None) -> np.ndarray:
    """Oracle ranking rule: score desc, index desc (see module docstring)."""
    order = np.argsort(x, kind="stable")[::-1]
    return order if k is None else order[:k]


def dense_passage_scores(passage_emb: np.ndarray, q: np.ndarray, exact: bool = True
                         ) -> Tuple[np.ndarray, np.ndarray]:
    """dense_passage_retrieval (HippoRAG.py:1467-1502 == StandardRAG.py:393-429):
    returns (sorted_doc_ids, sorted_doc_scores) of the min-max normalised scores."""
    s = _dot(passage_emb, q, exact)
    s = np.squeeze(s) if s.ndim == 2 else s
    s = min_max_normalize(s)
    ids = topk_desc(s)
    return ids, s[ids]


# --------------------------------------------------------------------------- #
# rerank_facts: HippoRAG.py:1659-1707
# --------------------------------------------------------------------------- #
def rerank_facts(query_fact_scores: np.ndarray, link_top_k: int,
                 filter_fn: Optional[Callable[[List[int]], List[int]]] = None
                 ) -> Tuple[List[int], List[int]]:
    """Top-``link_top_k`` candidate fact indices (all of them when F <= k,
    :1683-1688), then the LLM "recognition memory" filter (:1696,
    rerank.py:108-131) which returns a subset in its own order; identity when
    ``filter_fn`` is None.  Returns (candidates, kept)."""
    if len(query_fact_scores) == 0:
        return [], []
    cand = topk_desc(query_fact_scores, link_top_k).tolist()
    kept = list(cand) if filter_fn is None else list(filter_fn(list(cand)))
    return cand, kept[:link_top_k]


# --------------------------------------------------------------------------- #
# seed construction: HippoRAG.py:1574-1623 + get_top_k_weights :1505-1542
# --------------------------------------------------------------------------- #
def seed_weights(index: RefIndex, query_fact_scores: np.ndarray, kept_facts: Sequence[int],
                 link_top_k: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Entity ("phrase") part of the reset vector.

    Returns (vertex_ids int64 [m], weights float64 [m]), m <= link_top_k, in
    descending-weight order.  Raises AssertionError where the reference's
    ``assert np.count_nonzero(all_phrase_weights) == len(linking_score_map)``
    (:1541) would fire (a kept phrase whose weight is exactly 0).
    """
    if link_top_k is None:
        link_top_k = index.linking_top_k
    v = index.num_vertices
    phrase_weights = np.zeros(v)                       # :1577 (float64)
    number_of_occurs = np.zeros(v)                     # :1579
    order: List[int] = []
    for f in kept_facts:                               # :1583
        fact_score = query_fact_scores[f]              # :1587 (np.float32 scalar)
        for vid in (int(index.subj_vertex[f]), int(index.obj_vertex[f])):   # :1590
            if vid < 0:                                # :1595-1597 phrase not a node
                continue
            w = fact_score
            nc = int(index.num_chunks[vid])
            if nc > 0:                                 # :1600-1601
                w = w / nc                             # fp32 / python int -> fp32
            phrase_weights[vid] += w                   # :1603 (fp64 accumulate)
            number_of_occurs[vid] += 1                 # :1604
            if vid not in order:
                order.append(vid)
    phrase_weights = np.divide(phrase_weights, number_of_occurs,
                               out=np.zeros_like(phrase_weights),
                               where=number_of_occurs != 0)               # :1608
    items = [(vid, float(phrase_weights[vid])) for vid in order]          # :1610-1618
    if link_top_k:
        items = sorted(items, key=lambda t: t[1], reverse=True)[:link_top_k]   # :1528 (stable)
        kept_ids = {vid for vid, _ in items}
        nonzero = int(np.count_nonzero(phrase_weights[list(kept_ids)])) if kept_ids else 0
        # :1535-1541 every other vertex is zeroed, then the assert
        assert nonzero == len(items), "count_nonzero(all_phrase_weights) != len(linking_score_map)"
    else:
        items = sorted(items, key=lambda t: t[1], reverse=True)
    ids = np.array([t[0] for t in items], dtype=np.int64)
    wts = np.array([t[1] for t in items], dtype=np.float64)
    return ids, wts


def reset_vector(index: RefIndex, seed_ids: np.ndarray, seed_w: np.ndarray,
                 dpr_norm_by_passage: np.ndarray, passage_node_weight: Optional[float] = None
                 ) -> np.ndarray:
    """node_weights = phrase_weights + passage_weights (HippoRAG.py:1626-1638).

    ``dpr_norm_by_passage[p]`` is the min-max normalised DPR score of passage p
    (fp32); the second min_max_normalize at :1627 is the identity on it.  The
    product at :1633 is fp32 (np.float32 * python float), stored into a float64
    array."""
    if passage_node_weight is None:
        passage_node_weight = index.passage_node_weight
    node_weights = np.zeros(index.num_vertices)
    pw = dpr_norm_by_passage * passage_node_weight      # stays fp32 for fp32 input
    node_weights[index.passage_vertex] = pw
    node_weights[seed_ids] += seed_w
    assert node_weights.sum() > 0, "No phrases found in the graph for the given facts"  # :1644
    return node_weights


# --------------------------------------------------------------------------- #
# run_ppr: HippoRAG.py:1709-1749
# --------------------------------------------------------------------------- #
def run_ppr(index: RefIndex, reset_prob: np.ndarray, damping: Optional[float] = None,
            mode: str = "exact", iters: int = 20) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Returns (sorted_doc_ids, sorted_doc_scores, x) -- x is the whole PPR vector."""
    if damping is None:
        damping = 0.5                                   # :1734
    if mode == "exact":
        x = _ppr.ppr_exact(index.p, reset_prob, damping)
    elif mode == "power":
        x = _ppr.ppr_power(index.p, reset_prob, damping, iters)
    else:
        raise ValueError(mode)
    doc_scores = x[index.passage_vertex]               # :1745
    ids = topk_desc(doc_scores)                        # :1746 (+ oracle tie rule)
    return ids, doc_scores[ids], x


# --------------------------------------------------------------------------- #
# retrieve(): HippoRAG.py:459-480, one query
# --------------------------------------------------------------------------- #
@dataclass
class RefResult:
    fact_candidates: List[int] = field(default_factory=list)
    fact_candidate_scores: Optional[np.ndarray] = None
    kept_facts: List[int] = field(default_factory=list)
    seed_ids: Optional[np.ndarray] = None
    seed_w: Optional[np.ndarray] = None
    reset: Optional[np.ndarray] = None
    x: Optional[np.ndarray] = None
    used_dpr: bool = False
    sorted_doc_ids: Optional[np.ndarray] = None
    sorted_doc_scores: Optional[np.ndarray] = None


def retrieve_one(index: RefIndex, q_fact: np.ndarray, q_pass: np.ndarray,
                 filter_fn=None, exact_dot: bool = True, ppr_mode: str = "exact",
                 ppr_iters: int = 20) -> RefResult:
    """One iteration of the loop at HippoRAG.py:459-480."""
    res = RefResult()
    qfs = fact_scores(index.fact_emb, q_fact, exact_dot)                     # :461
    cand, kept = rerank_facts(qfs, index.linking_top_k, filter_fn)           # :462
    res.fact_candidates = cand
    res.fact_candidate_scores = qfs[cand] if len(cand) else np.array([], dtype=np.float32)
    res.kept_facts = kept
    dpr_ids, dpr_scores = dense_passage_scores(index.passage_emb, q_pass, exact_dot)
    if len(kept) == 0:                                                       # :467-469
        res.used_dpr = True
        res.sorted_doc_ids, res.sorted_doc_scores = dpr_ids, dpr_scores
        return res
    res.seed_ids, res.seed_w = seed_weights(index, qfs, kept)                # :1574-1623
    by_passage = np.empty_like(dpr_scores)
    by_passage[dpr_ids] = dpr_scores                                         # :1629-1633
    res.reset = reset_vector(index, res.seed_ids, res.seed_w, by_passage)    # :1638
    ids, scores, x = run_ppr(index, res.reset, index.damping, ppr_mode, ppr_iters)   # :1648
    res.x = x
    res.sorted_doc_ids, res.sorted_doc_scores = ids, scores
    return res


def retrieve_dpr_one(index: RefIndex, q_pass: np.ndarray, exact_dot: bool = True
                     ) -> Tuple[np.ndarray, np.ndarray]:
    """retrieve_dpr (HippoRAG.py:704-714) / StandardRAG.retrieve (:181-193)."""
    return dense_passage_scores(index.passage_emb, q_pass, exact_dot)
