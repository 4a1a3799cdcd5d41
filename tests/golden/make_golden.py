"""Generates the committed golden fixtures under tests/golden/ from the oracle.

    python tests/golden/make_golden.py

The reference holds NO golden vectors for this path and cannot be imported here
(igraph missing), so these fixtures pin the ORACLE (cross-checked against networkx / sparse solve in
tests/test_oracle_ppr.py at generation time, see the asserts below) and give the GPU tests inputs
that travel without the generator.  Inputs reused from the reference: the 9-document toy corpus and
its 3 queries (src/hipporag/utils/sample_data.py:1-11) with hand-written OpenIE triples, and the
MockEmbeddingModel recipe (tests/integration/run_vector_stores.py:31-44: deterministic unit vectors
per text) with md5 instead of Python's salted hash().
"""

import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import networkx as nx  # noqa: E402

import oracle  # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float, float_to_bf16_bits  # noqa: E402
from hipporag_amd.retriever import HippoRAG, RetrievalConfig  # noqa: E402

DOCS = [  # sample_data.py:1-11
    "Oliver Badman is a politician.",
    "George Rankin is a politician.",
    "Thomas Marwick is a politician.",
    "Cinderella attended the royal ball.",
    "The prince used the lost glass slipper to search the kingdom.",
    "When the slipper fit perfectly, Cinderella was reunited with the prince.",
    "Erik Hort's birthplace is Montebello.",
    "Marina is born in Minsk.",
    "Montebello is a part of Rockland County.",
]
QUERIES = ["What is George Rankin's occupation?", "How did Cinderella reach her happy ending?",
           "What county is Erik Hort's birthplace a part of?"]
TRIPLES = [  # hand-written OpenIE output (the LLM step is out of scope)
    [("Oliver Badman", "is", "politician")],
    [("George Rankin", "is", "politician")],
    [("Thomas Marwick", "is", "politician")],
    [("Cinderella", "attended", "royal ball")],
    [("prince", "used", "glass slipper"), ("prince", "searched", "kingdom")],
    [("glass slipper", "fit", "Cinderella"), ("Cinderella", "reunited with", "prince")],
    [("Erik Hort", "born in", "Montebello")],
    [("Marina", "born in", "Minsk")],
    [("Montebello", "part of", "Rockland County")],
]
DIM = 64


class MockEmbeddingModel:
    """Deterministic unit vectors per text; queries are pulled towards the words they share with
    indexed strings so that retrieval is meaningful."""

    def _vec(self, text):
        seed = int.from_bytes(hashlib.md5(text.encode()).digest()[:8], "little")
        v = np.random.Generator(np.random.PCG64(seed)).standard_normal(DIM)
        return v / np.linalg.norm(v)

    def batch_encode(self, texts, instruction=None, norm=True):
        if isinstance(texts, str):
            texts = [texts]
        out = []
        for t in texts:
            words = [w for w in "".join(c.lower() if c.isalnum() else " " for c in t).split() if len(w) > 2]
            v = 0.3 * self._vec(t) + sum(self._vec("w:" + w) for w in words)
            out.append(v / np.linalg.norm(v))
        return np.asarray(out, dtype=np.float32)


def toy_fixture():
    model = MockEmbeddingModel()
    rag = HippoRAG(RetrievalConfig(embedding_precision="bf16"), embedding_model=model)
    rag.index_from_openie(DOCS, TRIPLES)
    a = rag._arrays
    csr = a["csr"]
    v = csr.num_vertices
    rows = np.repeat(np.arange(v), np.diff(csr.row_ptr))
    # oracle index from the SAME edge semantics, built independently from the node_to_node_stats dict
    src, dst, w = [], [], []
    for (ka, kb), wt in rag.node_to_node_stats.items():
        src.append(rag.node_name_to_vertex_idx[ka]); dst.append(rag.node_name_to_vertex_idx[kb]); w.append(wt)
    p = oracle.column_normalize(oracle.build_symmetric_csr(v, src, dst, w))
    assert np.array_equal(p.indices, csr.col_idx) and np.allclose(p.data, csr.val, rtol=1e-7)
    index = oracle.RefIndex(bf16_bits_to_float(a["fact_emb"]), bf16_bits_to_float(a["passage_emb"]), a["subj"],
                            a["obj"], a["num_chunks"], a["passage_vertex"], p)
    qf = bf16_bits_to_float(float_to_bf16_bits(model.batch_encode(QUERIES, instruction="query_to_fact")))
    qp = bf16_bits_to_float(float_to_bf16_bits(model.batch_encode(QUERIES, instruction="query_to_passage")))
    out = {"src": np.array(src, np.int32), "dst": np.array(dst, np.int32), "w": np.array(w, np.float64),
           "row_ptr": csr.row_ptr, "col_idx": csr.col_idx, "val": csr.val, "rows": rows.astype(np.int32),
           "passage_vertex": a["passage_vertex"], "subj": a["subj"], "obj": a["obj"], "num_chunks": a["num_chunks"],
           "passage_emb_bits": a["passage_emb"], "fact_emb_bits": a["fact_emb"],
           "qf_bits": float_to_bf16_bits(qf), "qp_bits": float_to_bf16_bits(qp)}
    for q in range(len(QUERIES)):
        r = oracle.retrieve_one(index, qf[q], qp[q])
        assert not r.used_dpr
        out[f"q{q}_fact_candidates"] = np.array(r.fact_candidates, np.int32)
        out[f"q{q}_fact_scores"] = r.fact_candidate_scores
        out[f"q{q}_seed_ids"] = r.seed_ids.astype(np.int32)
        out[f"q{q}_seed_w"] = r.seed_w
        out[f"q{q}_reset"] = r.reset
        out[f"q{q}_x"] = r.x
        out[f"q{q}_doc_ids"] = r.sorted_doc_ids.astype(np.int32)
        out[f"q{q}_doc_scores"] = r.sorted_doc_scores
        # pin against networkx at generation time
        g = nx.MultiGraph(); g.add_nodes_from(range(v))
        for s, d, ww in zip(src, dst, w):
            if s != d:
                g.add_edge(s, d, weight=ww)
        pers = {i: float(r.reset[i]) for i in range(v)}
        pr = nx.pagerank(g, alpha=0.5, personalization=pers, dangling=pers, weight="weight", tol=1e-15, max_iter=1000)
        assert np.allclose(r.x, [pr[i] for i in range(v)], rtol=1e-9, atol=1e-14)
    np.savez_compressed(os.path.join(HERE, "toy_corpus.npz"), **out)
    print("toy_corpus.npz: V =", v, "nnz =", csr.nnz, "top docs:",
          [DOCS[int(out[f'q{q}_doc_ids'][0])] for q in range(3)])


def graph_fixtures():
    for n, m, seed in ((100, 420, 1), (200, 900, 2)):       # straddle PRPACK's 128-vertex switch
        rng = np.random.default_rng(seed)
        live = n - 4
        src = rng.integers(0, live, m); dst = rng.integers(0, live, m)
        w = rng.choice([1.0, 2.0, 4.0, 0.9], m)
        src = np.concatenate([src, dst[:50]]); dst = np.concatenate([dst, src[:50]]); w = np.concatenate([w, w[:50]])
        reset = np.zeros((3, n))
        for b in range(3):
            reset[b, rng.integers(0, live, 4)] = rng.random(4) + 0.2
            reset[b, live - 20:live] += 0.05 * rng.random(20).astype(np.float32)
        reset[1, n - 1] = 0.3                                  # seeded isolated vertex
        p = oracle.column_normalize(oracle.build_symmetric_csr(n, src, dst, w))
        x = np.stack([oracle.ppr_exact(p, reset[b], 0.5, "solve") for b in range(3)])
        x85 = np.stack([oracle.ppr_exact(p, reset[b], 0.85, "solve") for b in range(3)])
        np.savez_compressed(os.path.join(HERE, f"graph_{n}.npz"), n=n, src=src.astype(np.int32),
                            dst=dst.astype(np.int32), w=w, reset=reset, x_alpha050=x, x_alpha085=x85)
        print(f"graph_{n}.npz")


if __name__ == "__main__":
    toy_fixture()
    graph_fixtures()
