"""Index-time entity KNN on the MI355X: ``retrieve_knn`` with the reference's signature.

Mirrors reference src/hipporag/utils/embed_utils.py:6-94 (the only GPU compute the reference itself
issues: blocked ``torch.mm`` + ``torch.topk``), called by ``add_synonymy_edges``
(HippoRAG.py:959-1020) with k = synonymy_edge_topk = 2047.  Here: L2-normalise + hi/lo bf16 split
(csrc/knn.hip), three MFMA passes into an fp32 score block (csrc/sim_gemm.hip, ~fp32-accurate), exact
row top-k (csrc/topk.hip).  No key blocking is needed -- a [query_batch x n_keys] fp32 block fits
HBM comfortably -- so ``key_batch_size`` is accepted and ignored (the reference's per-block top-k
followed by a merge is exact, hence equivalent).

Ties: torch.topk leaves them unordered; here score desc, then larger key position first.
"""

from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import check


def retrieve_knn(query_ids: List[str], key_ids: List[str], query_vecs, key_vecs, k: int = 2047,
                 query_batch_size: int = 1000, key_batch_size: int = 10000, *, precision: str = "bf16x3",
                 return_arrays: bool = False):
    """Top-k keys per query by cosine similarity.

    Returns ``{query_id: (list of key ids, list of scores)}`` like the reference; with
    ``return_arrays=True`` returns ``(idx int32 [nq, k'], score fp32 [nq, k'])`` instead (k' =
    min(k, n_keys)), which is what a graph builder wants at scale.
    precision: "bf16x3" (three passes, |error| ~1e-6, default) or "bf16" (one pass, ~2e-3).
    """
    import torch
    if len(key_vecs) == 0:
        return {}                                                        # embed_utils.py:22
    if not torch.cuda.is_available():
        raise RuntimeError("retrieve_knn needs an MI355X-class GPU (no CPU fallback on this path)")
    if precision not in ("bf16x3", "bf16"):
        raise ValueError(precision)
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    kv = torch.as_tensor(np.asarray(key_vecs, dtype=np.float32)).to(dev).contiguous()
    n_keys, dim = kv.shape
    if dim % 8:
        raise ValueError("embedding dim must be a multiple of 8")
    kk = int(min(k, n_keys))
    if kk > 2048:
        raise ValueError("k > 2048 is not supported by the selection kernel")
    split = precision == "bf16x3"

    def norm_split(x):
        hi = torch.empty(x.shape, dtype=torch.int16, device=dev)
        lo = torch.empty(x.shape, dtype=torch.int16, device=dev) if split else None
        check(lib.hrag_normalize_split_bf16(x.data_ptr(), x.shape[0], x.shape[1], 1, hi.data_ptr(),
                                            lo.data_ptr() if split else None, stream))
        return hi, lo

    k_hi, k_lo = norm_split(kv)
    del kv
    qv_all = np.asarray(query_vecs, dtype=np.float32)
    nq = qv_all.shape[0]
    ld = (n_keys + 3) // 4 * 4
    out_idx = np.empty((nq, kk), np.int32)
    out_sc = np.empty((nq, kk), np.float32)
    qb = max(1, int(query_batch_size))
    scores = torch.empty((min(qb, nq), ld), dtype=torch.float32, device=dev)
    for lo_q in range(0, nq, qb):
        q = torch.from_numpy(qv_all[lo_q: lo_q + qb]).to(dev).contiguous()
        b = q.shape[0]
        q_hi, q_lo = norm_split(q)
        s = scores[:b]
        if split:   # small terms first
            check(lib.hrag_sim_gemm(k_hi.data_ptr(), n_keys, dim, q_lo.data_ptr(), b, s.data_ptr(), ld, 0, stream))
            check(lib.hrag_sim_gemm(k_lo.data_ptr(), n_keys, dim, q_hi.data_ptr(), b, s.data_ptr(), ld, 1, stream))
            check(lib.hrag_sim_gemm(k_hi.data_ptr(), n_keys, dim, q_hi.data_ptr(), b, s.data_ptr(), ld, 1, stream))
        else:
            check(lib.hrag_sim_gemm(k_hi.data_ptr(), n_keys, dim, q_hi.data_ptr(), b, s.data_ptr(), ld, 0, stream))
        idx = torch.empty((b, kk), dtype=torch.int32, device=dev)
        val = torch.empty((b, kk), dtype=torch.float32, device=dev)
        check(lib.hrag_topk_rows(s.data_ptr(), b, n_keys, ld, kk, 0, 0, idx.data_ptr(), val.data_ptr(),
                                 None, None, stream))
        out_idx[lo_q: lo_q + b] = idx.cpu().numpy()
        out_sc[lo_q: lo_q + b] = val.cpu().numpy()
    if return_arrays:
        return out_idx, out_sc
    results: Dict[str, Tuple[List[str], List[float]]] = {}
    for i in range(nq):                                                  # embed_utils.py:81-89
        results[query_ids[i]] = ([key_ids[j] for j in out_idx[i]], out_sc[i].tolist())
    return results


def synonymy_candidates(entity_keys: Sequence[str], entity_texts: Sequence[str], entity_embs, *,
                        topk: int = 2047, sim_threshold: float = 0.8, max_per_node: int = 100,
                        query_batch_size: int = 1000):
    """The selection loop of add_synonymy_edges (HippoRAG.py:992-1018) on top of retrieve_knn:
    for every entity with more than 2 alphanumeric characters, neighbours in score order until the
    score drops below the threshold or more than ``max_per_node`` have been taken (the reference's
    ``num_nns > 100`` check lets 101 through); self matches and empty phrases are skipped.
    Returns [(key_a, key_b, score)] ready for ``HippoRAG.index_from_openie(synonym_edges=...)``-style
    consumers working on keys."""
    import re
    idx, sc = retrieve_knn(list(entity_keys), list(entity_keys), entity_embs, entity_embs, k=topk,
                           query_batch_size=query_batch_size, return_arrays=True)
    edges = []
    for i, key in enumerate(entity_keys):
        if len(re.sub("[^A-Za-z0-9]", "", entity_texts[i])) <= 2:
            continue
        n = 0
        for j, s in zip(idx[i], sc[i]):
            if s < sim_threshold or n > max_per_node:
                break
            if j != i and entity_texts[j] != "":
                edges.append((key, entity_keys[j], float(s)))
                n += 1
    return edges
