"""Golden vectors for the INCREMENTAL life cycle of the index, produced by the REFERENCE'S OWN CODE
(tests/golden/ref_harness.py: /root/reference/src/hipporag with igraph / LLM / embedding model substituted):

    rag.index(docs A)  ->  rag.index(docs B, overlapping A)  ->  rag.delete(some docs)

    PYTHONHASHSEED=0 python tests/golden/make_ref_incremental.py      (re-execs itself with the seed)

Writes tests/golden/ref_incremental.npz.  After each of the three steps it records what the reference holds
(vertex names, the igraph edge list by NAME, the passage store order, the fact store contents, the
chunk-count divisor of every entity) and what retrieve() returns for a fixed query list.  Synonymy edges are
switched off (synonymy_edge_sim_threshold > 1): the index-time KNN is a separate row of SURVEY 8(f) and the
mirror class takes synonym edges as an explicit list.
"""

from __future__ import annotations

import os
import shutil
import sys
import tempfile

if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
import ref_harness as rh  # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float, float_to_bf16_bits  # noqa: E402


class Bf16Mock(mg.MockEmbeddingModel):
    def batch_encode(self, texts, instruction=None, norm=True):
        e = super().batch_encode(texts, instruction=instruction, norm=norm)
        return bf16_bits_to_float(float_to_bf16_bits(e)).astype(np.float32)


# documents 0..5 first, then 3..10 (3, 4, 5 are re-submitted: they must collapse; 9 and 10 are the two extra
# documents below), then 1 and 6 are deleted
STEP_A = list(range(0, 6))
DELETE = [1, 6]
EXTRA_DOCS = ["Ada Lovelace wrote the first published algorithm for the Analytical Engine.",
              "Charles Babbage designed the Analytical Engine in London."]
EXTRA_TRIPLES = [[("Ada Lovelace", "wrote", "first algorithm"), ("first algorithm", "written for", "Analytical Engine")],
                 [("Charles Babbage", "designed", "Analytical Engine"), ("Charles Babbage", "worked in", "London")]]


STEP_B = list(range(3, len(mg.DOCS) + len(EXTRA_DOCS)))


def corpus():
    docs = list(mg.DOCS) + EXTRA_DOCS
    triples = [list(t) for t in mg.TRIPLES] + EXTRA_TRIPLES
    return docs, triples


def snapshot(rag, queries, tag, out):
    # index() does not reset ready_to_retrieve (only delete() does, HippoRAG.py:411): the retrieval objects
    # have to be refreshed by hand after an incremental index(), as a user of the reference must
    rag.prepare_retrieval_objects()
    g = rag.graph
    names = [v["name"] for v in g.vs]
    es = g.get_edgelist()
    w = list(g.es["weight"]) if es else []
    out[f"{tag}_vertex_names"] = np.array(names)
    out[f"{tag}_edge_src_name"] = np.array([names[a] for a, _ in es])
    out[f"{tag}_edge_dst_name"] = np.array([names[b] for _, b in es])
    out[f"{tag}_edge_w"] = np.asarray(w, np.float64)
    out[f"{tag}_passage_keys"] = np.array(list(rag.passage_node_keys))
    out[f"{tag}_passage_texts"] = np.array([rag.chunk_embedding_store.get_row(k)["content"] for k in rag.passage_node_keys])
    rows = rag.fact_embedding_store.get_rows(list(rag.fact_node_keys)) if len(rag.fact_node_keys) else {}
    out[f"{tag}_fact_contents"] = np.array([rows[k]["content"] for k in rag.fact_node_keys])
    ent_keys = sorted(rag.ent_node_to_chunk_ids)
    out[f"{tag}_ent_keys"] = np.array(ent_keys)
    out[f"{tag}_ent_num_chunks"] = np.array([len(rag.ent_node_to_chunk_ids[k]) for k in ent_keys], np.int64)
    sols = rag.retrieve(list(queries), num_to_retrieve=5)
    out[f"{tag}_docs"] = np.array([[d for d in s.docs] + [""] * (5 - len(s.docs)) for s in sols])
    out[f"{tag}_scores"] = np.array([list(s.doc_scores) + [0.0] * (5 - len(s.docs)) for s in sols], np.float64)


def main(dst=None):
    assert rh.reference_available(), "needs /root/reference"
    docs, triples = corpus()
    queries = list(mg.QUERIES) + ["Who designed the Analytical Engine?"]
    tmp = tempfile.mkdtemp(prefix="refinc_")
    try:
        rag = rh.build_reference_rag(tmp, [docs[i] for i in STEP_A], [triples[i] for i in STEP_A], Bf16Mock(),
                                     synonymy_edge_sim_threshold=1.5)
        out = {"queries": np.array(queries)}
        snapshot(rag, queries, "a", out)
        # from here on the object behaves like a reference instance re-opened on its save_dir: the OpenIE
        # results on disk are reused (index() extracts the new chunks only, delete() finds the triples to drop)
        rag.global_config.force_openie_from_scratch = False
        rag.global_config.force_index_from_scratch = False
        rag.openie = rh.FixedOpenIE({d: t for d, t in zip(docs, triples)})
        rag.index([docs[i] for i in STEP_B])
        snapshot(rag, queries, "b", out)
        rag.delete([docs[i] for i in DELETE])
        snapshot(rag, queries, "c", out)
        np.savez_compressed(os.path.join(dst or HERE, "ref_incremental.npz"), **out)
        for t in "abc":
            print(t, "V", len(out[f"{t}_vertex_names"]), "igraph edges", len(out[f"{t}_edge_w"]), "passages",
                  len(out[f"{t}_passage_keys"]), "facts", len(out[f"{t}_fact_contents"]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
