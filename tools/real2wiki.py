"""The real-topology knowledge graph of tests/golden/real2wiki_triples.npz (made by tools/make_real2wiki.py from the
corpus the reference ships, /root/reference/reproduce/dataset/2wikimultihopqa_corpus.json: 6 119 passages, 40 789
entities, 136 512 triples -- integer ids only), in the two forms the repository consumes:

  openie_inputs()   documents + per-document triples (strings "e<id>") for the mirror's index_from_openie
                    (hipporag_amd.retriever.HippoRAG) -- the path a user of the reference takes;
  build_kg(tiles)   the engine-facing arrays built directly with numpy under the reference's graph rules (SURVEY 8 a10:
                    HippoRAG.py:867-957, :1159-1223): entity vertices (store order = sorted processed strings), then
                    passage vertices; a triple (s, p, o) of a chunk adds 1 to BOTH ordered pairs (s, o) and (o, s) -> the
                    two parallel igraph edges sum to weight 2 x count; passage -> entity edges weight 1; num_chunks[e] =
                    chunks that contain e; facts = distinct triples in first-occurrence order.  tests/test_real2wiki.py
                    checks that both forms give the SAME graph.

tiles > 1 (bench.py --config real2wiki): the corpus is replicated `tiles` times with disjoint entity names -- a stand-in
for a corpus `tiles` times larger with the same per-document topology.  Copies are DISCONNECTED (a real corpus of that
size would share its hub entities across all of it) and their ids are INTERLEAVED (entity id = local * tiles + tile:
what sorting all entity strings of the larger corpus would do), so the numbering as given has no locality to exploit;
graph.locality_order has to find it.  Embeddings follow the distribution of the reference's own mock recipe
(tests/integration/run_vector_stores.py:31-44: uniform [0, 1)^64, L2-normalised), drawn vectorised from one seeded
generator instead of one generator per string hash (Python's str hash is salted per process upstream)."""
from __future__ import annotations

import os

import numpy as np

FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real2wiki_triples.npz")
DIM = 64
PRED = ("mentions", "co-occurs with")


def load_fixture(max_passages=None):
    """(ptr, subj, pred, obj, n_entities, n_passages); max_passages: the first so many documents and only the entities
    they mention, renumbered in the same (sorted-string) order."""
    z = np.load(FIXTURE)
    ptr, subj, pred, obj = (z[k].astype(np.int64) for k in ("ptr", "subj", "pred", "obj"))
    n_e, n_p = int(z["n_entities"]), int(z["n_passages"])
    if max_passages is not None and max_passages < n_p:
        n_p = int(max_passages)
        subj, pred, obj, ptr = subj[:ptr[n_p]], pred[:ptr[n_p]], obj[:ptr[n_p]], ptr[:n_p + 1]
        used = np.unique(np.concatenate([subj, obj]))
        remap = np.full(n_e, -1, np.int64)
        remap[used] = np.arange(used.shape[0])
        subj, obj, n_e = remap[subj], remap[obj], int(used.shape[0])
    return ptr, subj, pred, obj, n_e, n_p


def openie_inputs(max_passages=None):
    """(docs, chunk_triples) for HippoRAG.index_from_openie: document i = "passage <i>", entity id -> "e<id, 6 digits>"
    (zero-padded: the mirror sorts entity STRINGS like the reference, which then is the id order)."""
    ptr, subj, pred, obj, n_e, n_p = load_fixture(max_passages)
    docs = [f"passage {i:06d}" for i in range(n_p)]
    triples = [[[f"e{subj[t]:06d}", PRED[pred[t]], f"e{obj[t]:06d}"] for t in range(ptr[i], ptr[i + 1])] for i in range(n_p)]
    return docs, triples


def mock_embeddings(rows: int, seed: int, dim: int = DIM) -> np.ndarray:
    """fp32 [rows, dim]: uniform [0, 1) L2-normalised (run_vector_stores.py:31-44's distribution), one seeded generator."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((rows, dim), np.float32)
    for lo in range(0, rows, 1 << 18):
        x = rng.random((min(1 << 18, rows - lo), dim), dtype=np.float32)
        out[lo:lo + x.shape[0]] = x / np.linalg.norm(x, axis=1, keepdims=True)
    return out


def build_kg(tiles: int = 1, interleave: bool = True, max_passages=None):
    """hipporag_amd.synth.SyntheticKG of the fixture, `tiles` disjoint copies (see the module docstring)."""
    from hipporag_amd.graph import build_csr
    from hipporag_amd.synth import SyntheticKG
    ptr, subj, pred, obj, n_e, n_p = load_fixture(max_passages)
    chunk = np.repeat(np.arange(n_p, dtype=np.int64), np.diff(ptr))
    # distinct triples in first-occurrence order = the fact store
    key = (subj * 2 + pred) * n_e + obj
    _, first = np.unique(key, return_index=True)
    first.sort()
    f_s, f_o = subj[first], obj[first]
    # fact edges: one count per (chunk, triple); both ordered pairs; the two parallel igraph edges sum
    lo, hi = np.minimum(subj, obj), np.maximum(subj, obj)
    pk, cnt = np.unique(lo * n_e + hi, return_counts=True)
    e_src, e_dst, e_w = pk // n_e, pk % n_e, 2.0 * cnt
    # passage -> entity edges: every entity of the chunk once
    ce = np.unique(np.concatenate([chunk * n_e + subj, chunk * n_e + obj]))
    p_idx, p_ent = ce // n_e, ce % n_e
    num_chunks = np.bincount(p_ent, minlength=n_e)
    T = int(tiles)
    ent_id = (lambda e, t: e * T + t) if interleave else (lambda e, t: t * n_e + e)
    pas_id = (lambda p, t: n_e * T + (p * T + t)) if interleave else (lambda p, t: n_e * T + t * n_p + p)
    src, dst, w, sv, ov = [], [], [], [], []
    nc = np.zeros((n_e + n_p) * T, np.int32)
    for t in range(T):
        src += [ent_id(e_src, t), pas_id(p_idx, t)]
        dst += [ent_id(e_dst, t), ent_id(p_ent, t)]
        w += [e_w, np.ones(p_idx.shape[0])]
        sv.append(ent_id(f_s, t)); ov.append(ent_id(f_o, t))
        nc[ent_id(np.arange(n_e), t)] = num_chunks
    src, dst, w = np.concatenate(src), np.concatenate(dst), np.concatenate(w)
    V = (n_e + n_p) * T
    # passage POSITION order (the chunk store's rows): tile-major documents, whatever their vertex ids
    pv = np.concatenate([pas_id(np.arange(n_p), t) for t in range(T)]).astype(np.int32)
    return SyntheticKG(num_vertices=V, n_entities=n_e * T, n_passages=n_p * T, src=src, dst=dst, weight=w,
                       csr=build_csr(V, src, dst, w), passage_vertex=pv, subj_vertex=np.concatenate(sv).astype(np.int32),
                       obj_vertex=np.concatenate(ov).astype(np.int32), num_chunks=nc)
