"""Working-directory loaders: parquet stores + OpenIE JSON in the reference's on-disk
format -> the same index arrays as indexing from memory (tests/golden/toy_corpus.npz)."""

import os

import numpy as np
import pytest

from hipporag_amd import RetrievalConfig
from hipporag_amd.loaders import (filter_invalid_triples, load_embedding_store, load_openie_results,
                                  load_reference_workdir, write_reference_workdir)
from tests.golden.make_golden import DOCS, QUERIES, TRIPLES, MockEmbeddingModel
from tests.helpers import tie_aware_equal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def workdir(tmp_path):
    save_dir = str(tmp_path / "outputs")
    write_reference_workdir(save_dir, "meta/llama-3", "nvidia/NV-Embed-v2", DOCS, TRIPLES, MockEmbeddingModel())
    return save_dir


def test_store_and_openie_files_round_trip(workdir):
    work = os.path.join(workdir, "meta_llama-3_nvidia_NV-Embed-v2")
    ids, texts, emb = load_embedding_store(os.path.join(work, "chunk_embeddings"), "chunk")
    assert texts == DOCS and emb.shape == (len(DOCS), 64) and emb.dtype == np.float32
    assert all(i.startswith("chunk-") for i in ids)
    np.testing.assert_allclose(emb, MockEmbeddingModel().batch_encode(DOCS), rtol=0, atol=0)
    assert load_embedding_store(os.path.join(work, "nope"), "chunk")[0] == []
    oi = load_openie_results(os.path.join(workdir, "openie_results_ner_meta_llama-3.json"))
    assert set(oi) == set(ids)
    assert filter_invalid_triples([["a", "b", "c"], ["a", "b"], ["a", "b", "c"], [1, 2, 3]]) == [["a", "b", "c"], ["1", "2", "3"]]


def test_workdir_loader_reproduces_toy_index(workdir):
    t = np.load(os.path.join(GOLD, "toy_corpus.npz"))
    rag = load_reference_workdir(workdir, "meta/llama-3", "nvidia/NV-Embed-v2", synonymy="none",
                                 embedding_model=MockEmbeddingModel(),
                                 global_config=RetrievalConfig(embedding_precision="bf16"))
    a = rag._arrays
    for key, arr in (("row_ptr", a["csr"].row_ptr), ("col_idx", a["csr"].col_idx), ("val", a["csr"].val),
                     ("subj", a["subj"]), ("obj", a["obj"]), ("num_chunks", a["num_chunks"]),
                     ("passage_vertex", a["passage_vertex"]), ("passage_emb_bits", a["passage_emb"]),
                     ("fact_emb_bits", a["fact_emb"])):
        np.testing.assert_array_equal(arr, t[key], err_msg=key)
    assert rag.passage_texts == DOCS and len(rag.facts) == len(t["subj"])


def test_workdir_loader_accepts_exported_igraph_edges(workdir, tmp_path):
    """graph_edges= (the export of graph.pickle): one row per igraph edge, parallel edges included."""
    rag0 = load_reference_workdir(workdir, "meta/llama-3", "nvidia/NV-Embed-v2", synonymy="none")
    names = rag0.entity_node_keys + rag0.passage_node_keys
    vid = rag0.node_name_to_vertex_idx
    src, dst, w = [], [], []
    for (a, b), wt in rag0.node_to_node_stats.items():
        src.append(vid[a]); dst.append(vid[b]); w.append(wt)
    perm = np.random.default_rng(0).permutation(len(names))            # igraph's own vertex numbering
    inv = np.empty_like(perm); inv[perm] = np.arange(len(names))
    path = str(tmp_path / "edges.npz")
    np.savez(path, names=np.asarray(names, dtype=str)[perm], src=inv[src], dst=inv[dst], weight=np.asarray(w))
    rag1 = load_reference_workdir(workdir, "meta/llama-3", "nvidia/NV-Embed-v2", graph_edges=path)
    for k in ("row_ptr", "col_idx", "val"):
        np.testing.assert_array_equal(getattr(rag1._arrays["csr"], k), getattr(rag0._arrays["csr"], k))


@pytest.mark.gpu
def test_gpu_retrieve_from_workdir(workdir, gpu_device):
    t = np.load(os.path.join(GOLD, "toy_corpus.npz"))
    rag = load_reference_workdir(workdir, "meta/llama-3", "nvidia/NV-Embed-v2", synonymy="none",
                                 embedding_model=MockEmbeddingModel(),
                                 global_config=RetrievalConfig(embedding_precision="bf16"))   # the fixture's vectors are bf16-rounded
    rag.global_config.max_batch, rag.global_config.ppr_iters = 4, 40
    sols = rag.retrieve(QUERIES, num_to_retrieve=5)
    for q, sol in enumerate(sols):
        got_ids = [DOCS.index(d) for d in sol.docs]
        assert tie_aware_equal(got_ids, t[f"q{q}_doc_ids"][:5], t[f"q{q}_doc_scores"][:5], rel_gap=2e-5)
        np.testing.assert_allclose(sol.doc_scores, t[f"q{q}_x"][t["passage_vertex"]][got_ids], rtol=1e-5)
    # with the synonymy edges recomputed on the GPU the graph only gains edges between entities
    rag2 = load_reference_workdir(workdir, "meta/llama-3", "nvidia/NV-Embed-v2", synonymy="knn",
                                  synonymy_edge_sim_threshold=0.3, embedding_model=MockEmbeddingModel())
    assert rag2._arrays["csr"].nnz >= rag._arrays["csr"].nnz
    assert len(rag2.retrieve(QUERIES[:1], num_to_retrieve=3)[0].docs) == 3
