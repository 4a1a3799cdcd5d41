"""Knowledge-graph -> CSR, with the reference's edge semantics.

The reference keeps an undirected igraph (``is_directed_graph=False``,
src/hipporag/utils/config_utils.py:176) filled from ``node_to_node_stats``
(src/hipporag/HippoRAG.py:1189-1223): every dict key becomes its OWN igraph edge, self pairs are
dropped (:1201); fact pairs are stored under both (s,o) and (o,s) (:906-910) and therefore end up
as two parallel edges.  PRPACK then treats every undirected edge as two directed ones and
normalises the weights per source vertex.  ``build_csr`` applies exactly that:

    for every edge (u, v, w), u != v:   A[u,v] += w ; A[v,u] += w
    P[i,j] = A[i,j] / sum_i A[i,j]        (column-stochastic; vertices without edges: dangling)

and returns P in CSR over OUTPUT vertices (row i lists its in-neighbours j) with int32 indices
and fp32 values (sums and the division in fp64, rounded once) -- the layout
``hrag_graph_desc`` (include/hrag.h) expects.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class CSRGraph:
    num_vertices: int
    row_ptr: np.ndarray   # int32 [V+1]
    col_idx: np.ndarray   # int32 [nnz]
    val: np.ndarray       # fp32  [nnz]  column-normalised
    raw: np.ndarray       # fp64  [nnz]  summed adjacency weights A[i,j] (before normalisation)
    col_sum: "np.ndarray | None" = None   # fp64 [V] weighted degree sum_i A[i,j] (hrag_graph_desc.col_sum)

    @property
    def nnz(self) -> int:
        return int(self.col_idx.shape[0])

    def rows(self, lo: int, hi: int) -> "CSRGraph":
        """Row shard [lo, hi) with global column ids (multi-GPU row sharding)."""
        a, b = int(self.row_ptr[lo]), int(self.row_ptr[hi])
        return CSRGraph(self.num_vertices, (self.row_ptr[lo:hi + 1] - a).astype(np.int32),
                        self.col_idx[a:b], self.val[a:b], self.raw[a:b], self.col_sum)


def looks_undirected(g: CSRGraph, rtol: float = 1e-5, samples: int = 4096, chunk: int = 1 << 22) -> bool:
    """Is the adjacency behind a column-normalised CSR symmetric (what HRAG_OPT_ACCEL's Chebyshev steps need: a real
    spectrum)?  Two tests on A_ij = P_ij d_j: (1) the ROW sums of A equal its column sums d (necessary; O(nnz), in
    chunks of `chunk` entries so that a 2e8-entry graph costs megabytes of temporaries, not gigabytes); (2) `samples`
    entries drawn at fixed strides have a mirror entry A_ji of the same weight (catches an asymmetric weighting with
    balanced sums).  build_csr symmetrises by construction, like the reference's undirected igraph (HippoRAG.py:236);
    this catches a directed graph handed to the C ABI.  Unsharded graphs with col_sum only."""
    if g.col_sum is None or g.row_ptr.shape[0] - 1 != g.num_vertices:
        return False
    d = np.asarray(g.col_sum, dtype=np.float64)
    nnz = int(g.nnz)
    if nnz == 0:
        return True
    rp = np.asarray(g.row_ptr, dtype=np.int64)
    col, val = np.asarray(g.col_idx), np.asarray(g.val)
    n = g.num_vertices
    r0 = 0
    while r0 < n:                      # whole rows per chunk
        r1 = int(np.searchsorted(rp, rp[r0] + chunk, side="right")) - 1
        r1 = min(max(r1, r0 + 1), n)
        lo, hi = int(rp[r0]), int(rp[r1])
        if hi > lo:
            a = val[lo:hi].astype(np.float64) * d[col[lo:hi]]
            starts = rp[r0:r1] - lo
            nz = rp[r0 + 1:r1 + 1] > rp[r0:r1]
            rs = np.zeros(r1 - r0)
            rs[nz] = np.add.reduceat(a, starts[nz])
            dd = d[r0:r1]
            if not np.all(np.abs(rs - dd) <= rtol * np.maximum(dd, 1e-300)):
                return False
        elif np.any(d[r0:r1] > 0):
            return False
        r0 = r1
    # sampled mirror entries: entry e = (i, j, A_ij) must meet (j, i, A_ji = A_ij)
    k = min(samples, nnz)
    e = (np.arange(k, dtype=np.int64) * (nnz // k)) if k else np.zeros(0, dtype=np.int64)
    i = np.searchsorted(rp, e, side="right") - 1
    j = col[e].astype(np.int64)
    aij = val[e].astype(np.float64) * d[j]
    for t in range(k):
        lo, hi = int(rp[j[t]]), int(rp[j[t] + 1])
        hit = np.flatnonzero(col[lo:hi] == i[t])
        if hit.size == 0:
            return False
        aji = float(val[lo + hit].astype(np.float64).sum()) * d[i[t]]
        if abs(aji - aij[t]) > 4 * rtol * max(abs(aij[t]), 1e-300):
            return False
    return True


def build_csr(num_vertices: int, src, dst, weight) -> CSRGraph:
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    w = np.asarray(weight, dtype=np.float64)
    if not (src.shape == dst.shape == w.shape):
        raise ValueError("src, dst, weight must have the same shape")
    if src.size and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= num_vertices):
        raise ValueError("edge endpoint outside [0, num_vertices)")
    keep = src != dst                       # HippoRAG.py:1201
    src, dst, w = src[keep], dst[keep], w[keep]
    rows = np.concatenate([src, dst])       # both directions of every undirected edge
    cols = np.concatenate([dst, src])
    vals = np.concatenate([w, w])
    key = rows * np.int64(num_vertices) + cols
    order = np.argsort(key, kind="stable")
    key, vals = key[order], vals[order]
    if key.size:
        first = np.concatenate([[True], key[1:] != key[:-1]])
        starts = np.flatnonzero(first)
        merged = np.add.reduceat(vals, starts)          # parallel edges sum
        ukey = key[starts]
    else:
        merged = np.zeros(0)
        ukey = key
    urows = ukey // num_vertices
    ucols = ukey % num_vertices
    if ukey.size >= 2**31 - 1:
        raise ValueError("nnz does not fit int32")
    colsum = np.bincount(ucols, weights=merged, minlength=num_vertices)
    val = (merged / colsum[ucols]).astype(np.float32) if ukey.size else np.zeros(0, np.float32)
    counts = np.bincount(urows, minlength=num_vertices)
    row_ptr = np.zeros(num_vertices + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    return CSRGraph(int(num_vertices), row_ptr.astype(np.int32), ucols.astype(np.int32), val, merged,
                    colsum.astype(np.float64))


def locality_order(csr: CSRGraph, passage_vertex) -> np.ndarray:
    """perm[old vertex] = new vertex: the graph compiler's locality numbering.

    The reference numbers its entity vertices in the order a Python set yields them (extract_entity_nodes,
    HippoRAG.py:1159-1187: hash order), so neighbouring ids have nothing to do with each other -- while the corpus
    has locality: a document's passages mention the same entities.  The rule here: non-passage vertices are ordered
    by the FIRST passage (in passage order) that links them (stable; vertices no passage links keep their relative
    order at the end), passage vertices follow in passage order.  Rows that are close in the new numbering then share
    in-neighbours, which is what hrag_opts.sell_sigma + HRAG_OPT_XCD_BLOCKED turn into L2 hits."""
    v = csr.num_vertices
    pv = np.asarray(passage_vertex, dtype=np.int64)
    is_p = np.zeros(v, dtype=bool)
    is_p[pv] = True
    first = np.full(v, np.iinfo(np.int64).max, dtype=np.int64)
    rp = np.asarray(csr.row_ptr, dtype=np.int64)
    lens = rp[pv + 1] - rp[pv]
    if lens.sum():
        # the entries of the passage rows, passage by passage
        starts = np.repeat(rp[pv], lens)
        offs = np.arange(lens.sum(), dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        cols = np.asarray(csr.col_idx, dtype=np.int64)[starts + offs]
        np.minimum.at(first, cols, np.repeat(np.arange(pv.shape[0], dtype=np.int64), lens))
    ents = np.flatnonzero(~is_p)
    order = ents[np.argsort(first[ents], kind="stable")]
    perm = np.empty(v, dtype=np.int64)
    perm[order] = np.arange(order.shape[0])
    perm[pv] = order.shape[0] + np.arange(pv.shape[0])
    return perm


def degree_order(csr: CSRGraph, passage_vertex) -> np.ndarray:
    """perm[old vertex] = new vertex: non-passage vertices by falling entry count (stable), passages after them in passage
    order.  A numbering for graphs WITHOUT corpus locality: with a narrow state (B <= 8: 2 ... 16 bytes per vertex, 8 ... 64
    vertices per cache line) the columns that most entries point at then share lines, which the per-CU cache can hold."""
    v = csr.num_vertices
    pv = np.asarray(passage_vertex, dtype=np.int64)
    is_p = np.zeros(v, dtype=bool)
    is_p[pv] = True
    deg = np.diff(np.asarray(csr.row_ptr, dtype=np.int64))
    ents = np.flatnonzero(~is_p)
    order = ents[np.argsort(-deg[ents], kind="stable")]
    perm = np.empty(v, dtype=np.int64)
    perm[order] = np.arange(order.shape[0])
    perm[pv] = order.shape[0] + np.arange(pv.shape[0])
    return perm


def relabel_csr(csr: CSRGraph, perm: np.ndarray) -> CSRGraph:
    """The same matrix with vertex i renamed perm[i] (rows and columns; columns stay sorted inside a row)."""
    v = csr.num_vertices
    perm = np.asarray(perm, dtype=np.int64)
    rows = np.repeat(np.arange(v, dtype=np.int64), np.diff(csr.row_ptr))
    new_r, new_c = perm[rows], perm[np.asarray(csr.col_idx, dtype=np.int64)]
    order = np.argsort(new_r * np.int64(v) + new_c, kind="stable")
    row_ptr = np.zeros(v + 1, dtype=np.int64)
    np.cumsum(np.bincount(new_r, minlength=v), out=row_ptr[1:])
    col_sum = None
    if csr.col_sum is not None:
        col_sum = np.empty(v, dtype=np.float64)
        col_sum[perm] = csr.col_sum
    return CSRGraph(v, row_ptr.astype(np.int32), new_c[order].astype(np.int32), np.asarray(csr.val)[order],
                    np.asarray(csr.raw)[order], col_sum)


def locality_score(csr: CSRGraph, window: int = 4096, perm: Optional[np.ndarray] = None) -> float:
    """Fraction of the matrix entries whose column lies within `window` ids of their row: how much a numbering gives
    the sweep to re-use (the benchmark generator: ~1 %; a corpus numbered by locality_order: most of them).
    perm: score the matrix as it WOULD be under that renumbering (no relabelled copy is built)."""
    rows = np.repeat(np.arange(csr.num_vertices, dtype=np.int64), np.diff(csr.row_ptr))
    if rows.size == 0:
        return 0.0
    cols = np.asarray(csr.col_idx, dtype=np.int64)
    if perm is not None:
        perm = np.asarray(perm, dtype=np.int64)
        rows, cols = perm[rows], perm[cols]
    return float(np.mean(np.abs(rows - cols) < window))


def float_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16, returned as uint16 (the wire format of hrag_embed_desc)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    rounded = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return rounded.astype(np.uint16)


def bf16_bits_to_float(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


class IncrementalGraph:
    """The reference's igraph object as this path needs it: named vertices in insertion order and an
    undirected MULTI-graph edge list (every add_edges call appends, parallel edges are legal and are summed
    by build_csr -- exactly what PRPACK sees, HippoRAG.py:1189-1223).  Incremental index() appends vertices
    and edges (:1159-1223); delete() removes vertices with igraph's renumbering (Graph.delete_vertices,
    :408): incident edges go, the remaining vertices keep their relative order and become 0..n-1."""

    def __init__(self):
        self.names: list = []
        self.index: dict = {}
        self._src = np.zeros(0, np.int64)
        self._dst = np.zeros(0, np.int64)
        self._w = np.zeros(0, np.float64)

    @property
    def num_vertices(self) -> int:
        return len(self.names)

    @property
    def num_edges(self) -> int:
        return int(self._src.shape[0])

    def edge_list(self):
        return self._src, self._dst, self._w

    def add_vertices(self, names) -> int:
        """Append the names that are not vertices yet (add_new_nodes :1159-1187); returns how many were new."""
        n0 = len(self.names)
        for nm in names:
            if nm not in self.index:
                self.index[nm] = len(self.names)
                self.names.append(nm)
        return len(self.names) - n0

    def add_edges(self, pairs, weights) -> int:
        """Append one edge per (source name, target name) whose endpoints both exist and differ
        (add_new_edges :1200-1223 drops self pairs and warns about unknown endpoints); returns the number added."""
        s, d, w = [], [], []
        for (a, b), wt in zip(pairs, weights):
            if a == b:
                continue
            ia, ib = self.index.get(a), self.index.get(b)
            if ia is None or ib is None:
                continue
            s.append(ia); d.append(ib); w.append(float(wt))
        self._src = np.concatenate([self._src, np.asarray(s, np.int64)])
        self._dst = np.concatenate([self._dst, np.asarray(d, np.int64)])
        self._w = np.concatenate([self._w, np.asarray(w, np.float64)])
        return len(s)

    def delete_vertices(self, names) -> np.ndarray:
        """Remove the named vertices and every edge touching them; returns old id -> new id (-1: deleted)."""
        n = len(self.names)
        gone = np.zeros(n, dtype=bool)
        for nm in names:
            i = self.index.get(nm)
            if i is not None:
                gone[i] = True
        remap = np.where(gone, -1, np.cumsum(~gone) - 1).astype(np.int64)
        keep_e = ~(gone[self._src] | gone[self._dst]) if self._src.size else np.zeros(0, bool)
        self._src, self._dst, self._w = remap[self._src[keep_e]], remap[self._dst[keep_e]], self._w[keep_e]
        self.names = [nm for nm, g in zip(self.names, gone) if not g]
        self.index = {nm: i for i, nm in enumerate(self.names)}
        return remap

    def to_csr(self) -> CSRGraph:
        return build_csr(len(self.names), self._src, self._dst, self._w)
