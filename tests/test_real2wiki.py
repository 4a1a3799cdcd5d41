"""The real-topology fixture (tests/golden/real2wiki_triples.npz, tools/make_real2wiki.py: a deterministic LLM-free triple
extractor over the corpus the reference ships) -- CPU side: the fixture is what its generator says, the direct numpy
graph builder (tools/real2wiki.build_kg, used by `bench.py --config real2wiki`) gives the SAME graph as the mirror's
index_from_openie (the reference's rules, HippoRAG.py:867-957, :1159-1223), tiling keeps the per-tile graph, and the
graph compiler's locality numbering finds the per-document locality a real corpus has."""

import os

import numpy as np
import pytest

from tools import real2wiki as rw


def test_fixture_shape():
    ptr, subj, pred, obj, n_e, n_p = rw.load_fixture()
    assert (n_p, n_e, subj.shape[0]) == (6119, 40789, 136512)
    assert ptr[0] == 0 and ptr[-1] == subj.shape[0] and np.all(np.diff(ptr) >= 0)
    assert subj.min() >= 0 and max(subj.max(), obj.max()) < n_e and np.all(subj != obj)
    assert set(np.unique(pred).tolist()) <= {0, 1}


def test_direct_builder_equals_the_mirror_on_the_first_documents():
    from hipporag_amd.retriever import HippoRAG, RetrievalConfig
    n = 300
    docs, triples = rw.openie_inputs(n)
    kg = rw.build_kg(1, max_passages=n)
    m = HippoRAG(RetrievalConfig(embedding_precision="bf16"))
    pe = rw.mock_embeddings(len(docs), 1)
    n_facts = len({tuple(t) for tr in triples for t in tr})
    m.index_from_openie(docs, triples, passage_embeddings=pe, fact_embeddings=rw.mock_embeddings(n_facts, 2))
    a = m._arrays
    assert a["csr"].num_vertices == kg.num_vertices and len(m.entity_node_keys) == kg.n_entities
    # the mirror's vertex order: entities in sorted-string order (= id order: zero-padded names), then documents
    np.testing.assert_array_equal(a["passage_vertex"], kg.passage_vertex)
    np.testing.assert_array_equal(a["csr"].row_ptr, kg.csr.row_ptr)
    np.testing.assert_array_equal(a["csr"].col_idx, kg.csr.col_idx)
    np.testing.assert_array_equal(a["csr"].raw, kg.csr.raw)
    np.testing.assert_array_equal(a["csr"].val, kg.csr.val)
    np.testing.assert_array_equal(a["subj"], kg.subj_vertex)
    np.testing.assert_array_equal(a["obj"], kg.obj_vertex)
    np.testing.assert_array_equal(a["num_chunks"], kg.num_chunks)


def test_tiles_are_disjoint_copies_with_interleaved_ids():
    one, four = rw.build_kg(1, max_passages=200), rw.build_kg(4, max_passages=200)
    assert four.num_vertices == 4 * one.num_vertices and four.csr.nnz == 4 * one.csr.nnz and four.n_facts == 4 * one.n_facts
    n_e = one.n_entities
    deg1, deg4 = np.diff(one.csr.row_ptr), np.diff(four.csr.row_ptr)
    for t in range(4):
        np.testing.assert_array_equal(deg4[np.arange(n_e) * 4 + t], deg1[:n_e])
    # an entity row of tile t only links vertices of tile t
    r = 5 * 4 + 2
    cols = four.csr.col_idx[four.csr.row_ptr[r]:four.csr.row_ptr[r + 1]]
    ent, pas = cols[cols < 4 * n_e], cols[cols >= 4 * n_e] - 4 * n_e
    assert np.all(ent % 4 == 2) and np.all(pas % 4 == 2)


def test_real_topology_has_locality_the_graph_compiler_finds():
    """What DESIGN 4.1 asserted and round 4's review asked to see measured: a real corpus' documents share entities with
    their neighbours in document order.  locality_score (share of the matrix entries whose column lies within 4096 ids
    of their row) of the graph as numbered by the reference's rule (sorted strings) vs after
    graph.locality_order; 16 interleaved tiles: the numbering as given has none, the compiler recovers the same score."""
    from hipporag_amd.graph import locality_order, locality_score
    kg = rw.build_kg(1)
    given = locality_score(kg.csr)
    found = locality_score(kg.csr, perm=locality_order(kg.csr, kg.passage_vertex))
    assert given < 0.2 and found > 0.3, (given, found)
    kg16 = rw.build_kg(16)
    g16 = locality_score(kg16.csr)
    f16 = locality_score(kg16.csr, perm=locality_order(kg16.csr, kg16.passage_vertex))
    assert g16 < 0.08 and abs(f16 - found) < 0.02, (g16, f16, found)


def test_loaders_round_trip_on_the_real_topology(tmp_path):
    """SURVEY 8 f3 on this graph: the first 250 documents written in the reference's on-disk format (parquet stores +
    OpenIE JSON) and read back give the arrays of indexing from memory."""
    from hipporag_amd import RetrievalConfig
    from hipporag_amd.loaders import load_reference_workdir, write_reference_workdir
    from hipporag_amd.retriever import HippoRAG
    from tests.golden.make_golden import MockEmbeddingModel
    docs, triples = rw.openie_inputs(250)
    save_dir = str(tmp_path / "outputs")
    write_reference_workdir(save_dir, "meta/llama-3", "nvidia/NV-Embed-v2", docs, triples, MockEmbeddingModel())
    rag = load_reference_workdir(save_dir, "meta/llama-3", "nvidia/NV-Embed-v2", synonymy="none",
                                 embedding_model=MockEmbeddingModel(), global_config=RetrievalConfig(embedding_precision="bf16"))
    mem = HippoRAG(RetrievalConfig(embedding_precision="bf16"), embedding_model=MockEmbeddingModel())
    mem.index_from_openie(docs, triples)
    a, b = rag._arrays, mem._arrays
    for k in ("row_ptr", "col_idx", "val", "raw"):
        np.testing.assert_array_equal(getattr(a["csr"], k), getattr(b["csr"], k), err_msg=k)
    for k in ("passage_vertex", "subj", "obj", "num_chunks", "passage_emb", "fact_emb"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    kg = rw.build_kg(1, max_passages=250)
    np.testing.assert_array_equal(a["csr"].col_idx, kg.csr.col_idx)
