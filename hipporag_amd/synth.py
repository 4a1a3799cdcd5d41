"""Seeded synthetic knowledge graphs / embeddings / queries for BASELINE.json's configs
(SURVEY.md section 8d).  Shapes mirror what HippoRAG.index() produces (reference
src/hipporag/HippoRAG.py:867-957, :1159-1223):

  vertices : N_e entity vertices first, N_p = V * passage_frac passage vertices last (:1171-1187)
  edges    : (1) passage-entity, weight 1.0 (:953): every passage links Poisson(8)+1 distinct
                 entities drawn with Zipf(0.6) popularity; entities nobody drew get one passage;
             (2) entity-entity fact edges on distinct pairs, co-occurrence count c in {1,2,3}
                 (p = .8/.15/.05) stored with the merged weight 2c that the reference's two parallel
                 igraph edges (s,o),(o,s) add up to (:906-910, :1220);
             (3) 5 % of the entity-entity edges are synonym edges, weight U[0.8, 1.0] (:1007-1018);
             the number of distinct undirected pairs is exactly E, so nnz = 2E after symmetrisation.
  facts    : F = N_e triples whose (subject, object) are endpoints of sampled entity-entity edges;
             num_chunks[e] = number of passages entity e occurs in (>= 1).
  embeddings: unit-norm Gaussian rows rounded to bf16; queries = normalise(row + 0.5 * unit noise).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .graph import CSRGraph, build_csr, float_to_bf16_bits, bf16_bits_to_float


@dataclass
class SyntheticKG:
    num_vertices: int
    n_entities: int
    n_passages: int
    src: np.ndarray
    dst: np.ndarray
    weight: np.ndarray
    csr: CSRGraph
    passage_vertex: np.ndarray   # int32 [Np]
    subj_vertex: np.ndarray      # int32 [F]
    obj_vertex: np.ndarray       # int32 [F]
    num_chunks: np.ndarray       # int32 [V]

    @property
    def n_facts(self) -> int:
        return int(self.subj_vertex.shape[0])


def _zipf_ranks(rng: np.random.Generator, n_items: int, size: int, s: float) -> np.ndarray:
    """Ranks in [0, n_items) with P(rank) ~ (rank+1)^-s (continuous inverse-CDF approximation)."""
    u = rng.random(size)
    x = u ** (1.0 / (1.0 - s))
    return np.minimum((x * n_items).astype(np.int64), n_items - 1)


def make_kg(num_vertices: int, num_edges: int, seed: int, passage_frac: float = 0.125,
            power_law: bool = False, zipf_s: float = 0.6, community: int = 0, local_frac: float = 0.9) -> SyntheticKG:
    """community > 0: a NON-BASELINE variant with locality (what indexing a corpus document by document produces,
    HippoRAG.py:867-957: a document's passages talk about the same entities, facts link entities of the same
    documents): entities come in communities of `community` consecutive ids, passage p belongs to community
    p * n_comm / N_p, `local_frac` of a passage's entities and of the entity-entity edges stay inside the community
    (Zipf popularity inside it), the rest is drawn globally as in the baseline generator."""
    if community > 0:
        return _make_kg_local(num_vertices, num_edges, seed, passage_frac, zipf_s, community, local_frac)
    rng = np.random.Generator(np.random.PCG64(seed))
    n_p = max(1, int(round(num_vertices * passage_frac)))
    n_e = num_vertices - n_p
    if n_e < 2:
        raise ValueError("need at least two entity vertices")
    perm = rng.permutation(n_e)            # popularity rank -> entity id

    # (1) passage - entity edges
    deg = rng.poisson(8.0, n_p) + 1
    deg = np.minimum(deg, n_e)
    p_rep = np.repeat(np.arange(n_p, dtype=np.int64), deg)
    ent = perm[_zipf_ranks(rng, n_e, p_rep.size, zipf_s)]
    pe = np.unique(p_rep * n_e + ent)      # distinct (passage, entity) pairs
    p_of, e_of = pe // n_e, pe % n_e
    covered = np.zeros(n_e, dtype=bool)
    covered[e_of] = True
    lonely = np.flatnonzero(~covered)
    if lonely.size:
        p_of = np.concatenate([p_of, rng.integers(0, n_p, lonely.size)])
        e_of = np.concatenate([e_of, lonely])
    n_pe = p_of.size
    # (2)+(3) entity - entity edges on distinct pairs
    n_ee = max(0, num_edges - n_pe)
    lo = np.zeros(0, dtype=np.int64)
    hi = np.zeros(0, dtype=np.int64)
    keys = np.zeros(0, dtype=np.int64)
    max_pairs = n_e * (n_e - 1) // 2
    n_ee = min(n_ee, max_pairs)
    while keys.size < n_ee:
        need = int((n_ee - keys.size) * 1.3) + 16
        a = perm[_zipf_ranks(rng, n_e, need, zipf_s)]
        b = perm[_zipf_ranks(rng, n_e, need, zipf_s)] if power_law else rng.integers(0, n_e, need)
        ok = a != b
        a, b = a[ok], b[ok]
        k = np.minimum(a, b) * n_e + np.maximum(a, b)
        keys = np.unique(np.concatenate([keys, k]))
    if keys.size > n_ee:
        keys = rng.permutation(keys)[:n_ee]
    lo, hi = keys // n_e, keys % n_e
    c = rng.choice(np.array([1.0, 2.0, 3.0]), size=n_ee, p=[0.8, 0.15, 0.05])
    w_ee = 2.0 * c
    syn = rng.random(n_ee) < 0.05
    w_ee[syn] = rng.uniform(0.8, 1.0, int(syn.sum()))

    src = np.concatenate([n_e + p_of, lo])
    dst = np.concatenate([e_of, hi])
    weight = np.concatenate([np.ones(n_pe), w_ee])
    csr = build_csr(num_vertices, src, dst, weight)

    n_f = n_e
    if n_ee > 0:
        pick = rng.integers(0, n_ee, n_f)
        flip = rng.random(n_f) < 0.5
        subj = np.where(flip, lo[pick], hi[pick])
        obj = np.where(flip, hi[pick], lo[pick])
    else:  # degenerate tiny graphs: facts between random entities
        subj = rng.integers(0, n_e, n_f)
        obj = rng.integers(0, n_e, n_f)
    num_chunks = np.zeros(num_vertices, dtype=np.int32)
    num_chunks[:n_e] = np.bincount(e_of, minlength=n_e)
    return SyntheticKG(num_vertices, n_e, n_p, src, dst, weight, csr,
                       (n_e + np.arange(n_p)).astype(np.int32),
                       subj.astype(np.int32), obj.astype(np.int32), num_chunks)


def _make_kg_local(num_vertices, num_edges, seed, passage_frac, zipf_s, community, local_frac) -> SyntheticKG:
    rng = np.random.Generator(np.random.PCG64(seed))
    n_p = max(1, int(round(num_vertices * passage_frac)))
    n_e = num_vertices - n_p
    n_comm = max(1, n_e // community)
    comm_lo = (np.arange(n_comm, dtype=np.int64) * n_e) // n_comm          # first entity of a community
    comm_sz = np.diff(np.append(comm_lo, n_e))

    def draw(comm, size_like):      # entities: local_frac inside `comm` (Zipf rank inside it), the rest global
        local = rng.random(size_like.shape[0]) < local_frac
        r = _zipf_ranks(rng, 1 << 20, size_like.shape[0], zipf_s) / float(1 << 20)
        inside = comm_lo[comm] + np.minimum((r * comm_sz[comm]).astype(np.int64), comm_sz[comm] - 1)
        return np.where(local, inside, rng.integers(0, n_e, size_like.shape[0]))

    # (1) passage - entity edges
    deg = np.minimum(rng.poisson(8.0, n_p) + 1, n_e)
    p_rep = np.repeat(np.arange(n_p, dtype=np.int64), deg)
    p_comm = (p_rep * n_comm) // n_p
    ent = draw(p_comm, p_rep)
    pe = np.unique(p_rep * n_e + ent)
    p_of, e_of = pe // n_e, pe % n_e
    covered = np.zeros(n_e, dtype=bool)
    covered[e_of] = True
    lonely = np.flatnonzero(~covered)
    if lonely.size:         # an entity nobody drew gets a passage of its own community
        c = np.searchsorted(comm_lo, lonely, side="right") - 1
        p_lo, p_hi = (c * n_p) // n_comm, np.maximum(((c + 1) * n_p) // n_comm, (c * n_p) // n_comm + 1)
        p_of = np.concatenate([p_of, np.minimum(p_lo + (rng.random(lonely.size) * (p_hi - p_lo)).astype(np.int64), n_p - 1)])
        e_of = np.concatenate([e_of, lonely])
    n_pe = p_of.size
    # (2)+(3) entity - entity edges on distinct pairs
    n_ee = min(max(0, num_edges - n_pe), n_e * (n_e - 1) // 2)
    keys = np.zeros(0, dtype=np.int64)
    while keys.size < n_ee:
        need = int((n_ee - keys.size) * 1.3) + 16
        a = rng.integers(0, n_e, need)
        b = draw(np.searchsorted(comm_lo, a, side="right") - 1, a)
        ok = a != b
        a, b = a[ok], b[ok]
        keys = np.unique(np.concatenate([keys, np.minimum(a, b) * n_e + np.maximum(a, b)]))
    if keys.size > n_ee:
        keys = rng.permutation(keys)[:n_ee]
    lo, hi = keys // n_e, keys % n_e
    w_ee = 2.0 * rng.choice(np.array([1.0, 2.0, 3.0]), size=n_ee, p=[0.8, 0.15, 0.05])
    syn = rng.random(n_ee) < 0.05
    w_ee[syn] = rng.uniform(0.8, 1.0, int(syn.sum()))
    src = np.concatenate([n_e + p_of, lo])
    dst = np.concatenate([e_of, hi])
    weight = np.concatenate([np.ones(n_pe), w_ee])
    csr = build_csr(num_vertices, src, dst, weight)
    n_f = n_e
    pick = rng.integers(0, max(n_ee, 1), n_f)
    flip = rng.random(n_f) < 0.5
    subj = np.where(flip, lo[pick], hi[pick]) if n_ee else rng.integers(0, n_e, n_f)
    obj = np.where(flip, hi[pick], lo[pick]) if n_ee else rng.integers(0, n_e, n_f)
    num_chunks = np.zeros(num_vertices, dtype=np.int32)
    num_chunks[:n_e] = np.bincount(e_of, minlength=n_e)
    return SyntheticKG(num_vertices, n_e, n_p, src, dst, weight, csr, (n_e + np.arange(n_p)).astype(np.int32),
                       subj.astype(np.int32), obj.astype(np.int32), num_chunks)


def hash_order(kg: SyntheticKG, seed: int) -> SyntheticKG:
    """The same graph with its ENTITY vertex ids shuffled: the numbering the reference produces (entity vertices are
    added in the order a Python set yields them, HippoRAG.py:1159-1187), under which no locality of the corpus shows
    in the ids.  What graph.locality_order is there to undo."""
    from .graph import relabel_csr
    rng = np.random.Generator(np.random.PCG64(seed))
    h = np.arange(kg.num_vertices, dtype=np.int64)
    h[:kg.n_entities] = rng.permutation(kg.n_entities)
    nc = np.zeros_like(kg.num_chunks)
    nc[h] = kg.num_chunks
    return SyntheticKG(kg.num_vertices, kg.n_entities, kg.n_passages, h[kg.src], h[kg.dst], kg.weight,
                       relabel_csr(kg.csr, h), kg.passage_vertex, h[kg.subj_vertex].astype(np.int32),
                       h[kg.obj_vertex].astype(np.int32), nc)


def make_embeddings_np(rows: int, dim: int, seed: int) -> np.ndarray:
    """Unit-norm Gaussian rows rounded to bf16; returns the uint16 bit patterns [rows, dim]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((rows, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return float_to_bf16_bits(x)


def make_queries_np(emb_bits: np.ndarray, batch: int, seed: int, noise: float = 0.5
                    ) -> Tuple[np.ndarray, np.ndarray]:
    """Queries near randomly chosen rows (so that the top hit is well separated).
    Returns (uint16 bf16 bits [batch, dim], chosen row ids)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rows, dim = emb_bits.shape
    pick = rng.integers(0, rows, batch)
    base = bf16_bits_to_float(emb_bits[pick])
    nz = rng.standard_normal((batch, dim), dtype=np.float32)
    nz /= np.linalg.norm(nz, axis=1, keepdims=True)
    q = base + noise * nz
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return float_to_bf16_bits(q), pick


def make_embeddings_torch(rows: int, dim: int, seed: int, device, chunk: int = 1 << 18, dtype=None):
    """Same recipe generated on the device (bench-scale matrices): bf16 (default) or fp16 tensor [rows, dim]."""
    import torch
    dtype = dtype or torch.bfloat16
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((rows, dim), dtype=dtype, device=device)
    for r0 in range(0, rows, chunk):
        r1 = min(rows, r0 + chunk)
        x = torch.randn((r1 - r0, dim), generator=g, device=device, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        out[r0:r1] = x.to(dtype)
    return out


def make_queries_torch(emb, batch: int, seed: int, noise: float = 0.5):
    import torch
    g = torch.Generator(device=emb.device)
    g.manual_seed(seed)
    pick = torch.randint(0, emb.shape[0], (batch,), generator=g, device=emb.device)
    base = emb[pick].float()
    nz = torch.randn(base.shape, generator=g, device=emb.device, dtype=torch.float32)
    nz /= nz.norm(dim=1, keepdim=True)
    q = base + noise * nz
    q /= q.norm(dim=1, keepdim=True)
    return q.to(emb.dtype if emb.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16), pick
