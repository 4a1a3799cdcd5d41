"""Index-time entity KNN on the MI355X: ``retrieve_knn`` with the reference's signature.

Mirrors reference src/hipporag/utils/embed_utils.py:6-94 (the only GPU compute the reference itself
issues: blocked ``torch.mm`` + ``torch.topk``), called by ``add_synonymy_edges``
(HippoRAG.py:959-1020) with k = synonymy_edge_topk = 2047.  Here: L2-normalise + split into two fp16 halves
in the layout [hi | lo | hi] (keys) / [hi | hi | lo] (queries) (csrc/knn.hip ``split3_kernel``), ONE pass of the
wide-batch 256-row MFMA GEMM over 3 * dim elements into an fp32 score block (every product exact, the sum is the
fp32 product to 2^-21: ``precision="f32"``, the default), exact row top-k (csrc/topk.hip).  Round 1 ran three
accumulating passes of bf16 halves on the 128-row kernel (61 TFLOP/s effective).  No key blocking is needed -- a [query_batch x n_keys] fp32 block fits
HBM comfortably -- so ``key_batch_size`` is accepted and ignored (the reference's per-block top-k
followed by a merge is exact, hence equivalent).

Ties: torch.topk leaves them unordered; here score desc, then larger key position first.
"""

from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import check

# bound on |x . q - hi(x) . hi(q)| for unit vectors in the split layout: |lo| <= 2^-11 |x| per vector, two cross terms
# (9.8e-4) + the dropped lo . lo term and the fp32 accumulation (< 1e-5): what the thresholded first pass may miss by
_PREFIX_MARGIN = 1.2e-3


def retrieve_knn(query_ids: List[str], key_ids: List[str], query_vecs, key_vecs, k: int = 2047,
                 query_batch_size: int = 1000, key_batch_size: int = 10000, *, precision: str = "f32",
                 return_arrays: bool = False, min_score: float = None):
    """Top-k keys per query by cosine similarity.

    Returns ``{query_id: (list of key ids, list of scores)}`` like the reference; with
    ``return_arrays=True`` returns ``(idx int32 [nq, k'], score fp32 [nq, k'])`` instead (k' =
    min(k, n_keys)), which is what a graph builder wants at scale.
    min_score: the caller only reads neighbours down to this score (add_synonymy_edges stops at the first score
    below synonymy_edge_sim_threshold, HippoRAG.py:1004-1007): the device result is cut to the longest row prefix that
    reaches it before it crosses PCIe (entries below it come back as idx -1 / score 0) -- at k = 2047 the full arrays
    are 16 KB per query and dominate the call.
    precision: "f32" (fp16 hi + lo halves, one pass over 3 * dim: |error| ~1e-7, default; "bf16x3" is accepted as an
    alias) or "bf16" (rounded vectors, one pass over dim, ~2e-3).
    """
    import torch
    if len(key_vecs) == 0:
        return {}                                                        # embed_utils.py:22
    if not torch.cuda.is_available():
        raise RuntimeError("retrieve_knn needs an MI355X-class GPU (no CPU fallback on this path)")
    if precision not in ("f32", "bf16x3", "bf16"):
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
    split = precision != "bf16"
    kdim, dtype = (3 * dim, 1) if split else (dim, 0)        # hrag_dtype: fp16 halves / bf16

    def prepare(x, as_query):
        """F.normalize + the kernel's input layout"""
        if split:
            out = torch.empty((x.shape[0], 3 * dim), dtype=torch.int16, device=dev)
            check(lib.hrag_split_f32(x.data_ptr(), x.shape[0], dim, 1 if as_query else 0, 1, out.data_ptr(), stream))
            return out
        hi = torch.empty(x.shape, dtype=torch.int16, device=dev)
        check(lib.hrag_normalize_split_bf16(x.data_ptr(), x.shape[0], dim, 1, hi.data_ptr(), None, stream))
        return hi

    keys = prepare(kv, False)
    del kv
    qv_all = np.asarray(query_vecs, dtype=np.float32)
    nq = qv_all.shape[0]
    ld = (n_keys + 3) // 4 * 4
    out_idx = np.full((nq, kk), -1, np.int32) if min_score is not None else np.empty((nq, kk), np.int32)
    out_sc = np.zeros((nq, kk), np.float32) if min_score is not None else np.empty((nq, kk), np.float32)
    qb = max(1, int(query_batch_size))
    if min_score is not None:
        qb = max(qb, 4096)     # no [qb, n_keys] block in this mode: wider query tiles keep the key stream on the MFMAs
    scores = None
    fused_k = min(16, n_keys)
    if min_score is not None:
        # the thresholded call never writes a [qb, n_keys] block unless a query has 16 or more neighbours above the
        # threshold: the fused top-16 (tile maxima + rescore, exact) answers the others
        ws_bytes = int(lib.hrag_sim_topk_workspace_bytes(n_keys, min(qb, nq)))
        ws = torch.zeros((ws_bytes,), dtype=torch.uint8, device=dev)

    def dense(qq, rows_out):
        """[b, n_keys] score block + exact row top-k for the prepared queries qq; rows_out: their rows of the result
        (thresholded mode: cut to the prefix above min_score)"""
        nonlocal scores
        b = qq.shape[0]
        if scores is None or scores.shape[0] < b:
            scores = torch.empty((b, ld), dtype=torch.float32, device=dev)
        s = scores[:b]
        check(lib.hrag_sim_gemm(keys.data_ptr(), n_keys, kdim, qq.data_ptr(), b, s.data_ptr(), ld, 0, dtype, stream))
        idx = torch.empty((b, kk), dtype=torch.int32, device=dev)
        val = torch.empty((b, kk), dtype=torch.float32, device=dev)
        check(lib.hrag_topk_rows(s.data_ptr(), b, n_keys, ld, kk, 0, 0, idx.data_ptr(), val.data_ptr(),
                                 None, None, stream))
        if min_score is not None:      # rows are sorted: keep the longest prefix any of these rows needs
            keep = val >= float(min_score)
            w = int(keep.sum(1).max().item())
            out_idx[rows_out, :] = -1
            out_sc[rows_out, :] = 0
            if w:
                out_idx[rows_out, :w] = torch.where(keep[:, :w], idx[:, :w], torch.full_like(idx[:, :w], -1)).cpu().numpy()
                out_sc[rows_out, :w] = torch.where(keep[:, :w], val[:, :w], torch.zeros_like(val[:, :w])).cpu().numpy()
            return
        out_idx[rows_out] = idx.cpu().numpy()
        out_sc[rows_out] = val.cpu().numpy()

    if min_score is not None:
        # The thresholded fused top-16 (include/hrag.h hrag_sim_topk_min_score): tiles below the threshold are never
        # rescored, and on the split layout the tile maxima come from the hi . qhi third alone -- the other two thirds
        # move a score by at most 2 * 2^-11 = 9.8e-4 for unit vectors (margin 1.2e-3) -- a third of the MFMA work; what
        # is rescored is rescored over all 3 * dim elements, so every returned score is the exact chain.  Every block is
        # ENQUEUED before anything is read back (16 candidates per query stay on the device: 128 bytes each), so the
        # device never waits for the host between blocks (round 6: the per-block read-back cost 20 % of the call).
        i16 = torch.empty((nq, fused_k), dtype=torch.int32, device=dev)
        v16 = torch.empty((nq, fused_k), dtype=torch.float32, device=dev)
        over = torch.empty((nq,), dtype=torch.int32, device=dev)
        for lo_q in range(0, nq, qb):
            q = torch.from_numpy(qv_all[lo_q: lo_q + qb]).to(dev).contiguous()
            b = q.shape[0]
            qq = prepare(q, True)
            check(lib.hrag_sim_topk_min_score(keys.data_ptr(), n_keys, kdim, qq.data_ptr(), b, fused_k, dtype,
                                              dim if split else 0, float(min_score), _PREFIX_MARGIN if split else 0.0,
                                              ws.data_ptr(), ws_bytes, i16[lo_q: lo_q + b].data_ptr(),
                                              v16[lo_q: lo_q + b].data_ptr(), over[lo_q: lo_q + b].data_ptr(), stream))
        keep = v16 >= float(min_score)
        w = min(int(keep.sum(1).max().item()), kk)
        if w:
            out_idx[:, :w] = torch.where(keep[:, :w], i16[:, :w], torch.full_like(i16[:, :w], -1)).cpu().numpy()
            out_sc[:, :w] = torch.where(keep[:, :w], v16[:, :w], torch.zeros_like(v16[:, :w])).cpu().numpy()
        # the dense path for the queries the fused form cannot answer: more than 16 tiles that reach the threshold, or a
        # 16th neighbour still above it
        need = over != 0
        if kk > fused_k and n_keys > fused_k:
            need = need | keep[:, fused_k - 1]
        more = torch.nonzero(need).flatten().cpu().numpy()
        db = max(1, min(1024, int(query_batch_size)))
        for lo_m in range(0, len(more), db):
            rows_out = more[lo_m: lo_m + db]
            dense(prepare(torch.from_numpy(qv_all[rows_out]).to(dev).contiguous(), True), rows_out)
    else:
        for lo_q in range(0, nq, qb):
            q = torch.from_numpy(qv_all[lo_q: lo_q + qb]).to(dev).contiguous()
            dense(prepare(q, True), slice(lo_q, lo_q + q.shape[0]))
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
    # the loop below reads at most max_per_node + 1 neighbours (+ the self match) and stops at the threshold: the
    # same edges as from the full top-`topk` lists, without selecting / shipping 2047 entries per entity
    idx, sc = retrieve_knn(list(entity_keys), list(entity_keys), entity_embs, entity_embs,
                           k=min(topk, max_per_node + 3), query_batch_size=query_batch_size, return_arrays=True,
                           min_score=sim_threshold)
    edges = []
    for i, key in enumerate(entity_keys):
        if len(re.sub("[^A-Za-z0-9]", "", entity_texts[i])) <= 2:
            continue
        n = 0
        for j, s in zip(idx[i], sc[i]):
            if j < 0 or s < sim_threshold or n > max_per_node:
                break
            if j != i and entity_texts[j] != "":
                edges.append((key, entity_keys[j], float(s)))
                n += 1
    return edges
