"""An indexed reference ``hipporag.HippoRAG`` object restored from its recorded state (tests/golden/ref_state_*.pkl,
written by tests/golden/make_ref_golden.py from a REAL reference object after the reference's own index() and
prepare_retrieval_objects(), HippoRAG.py:1287-1389).  The reference package does not travel to the GPU box; its
object's state does: every attribute below holds what the real class computed -- vertex numbering, igraph edge list
with its parallel edges, store rows, md5 keys, fp32 embedding matrices, the query embedding cache -- so that
``reference_adapter.attach()`` meets the real engine on real reference state (tests/test_gpu_adapter_ref_state.py).

Unlike tests/fake_reference.py nothing here is derived with this repo's own graph rules."""

from __future__ import annotations

import os
import pickle
import types

from tests.fake_reference import FakeIGraph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class RowStore:
    """embedding_store.EmbeddingStore's read side (get_row / get_rows / get_all_ids), rows as recorded."""

    def __init__(self, rows):
        self._rows = dict(rows)

    def get_all_ids(self):
        return list(self._rows)

    def get_row(self, key):
        return self._rows[key]

    def get_rows(self, keys):
        return {k: self._rows[k] for k in keys}


def recorded_filter(queries, mode):
    """The stand-in for the LLM filter the generator ran the reference with (make_ref_golden.make_filter): a SUBSET
    of the candidates in ITS OWN order -- identity / reorder + subset / drop everything (DPR fallback)."""

    def filt(query, candidate_items, candidate_indices, len_after_rerank=None):
        qi = queries.index(query)
        how = "identity" if mode == "identity" else ("identity", "subset", "identity", "none", "subset")[qi % 5]
        if how == "identity":
            keep = list(range(len(candidate_indices)))
        elif how == "subset":
            keep = [i for i in (2, 0, 3) if i < len(candidate_indices)]
        else:
            keep = []
        return [candidate_indices[i] for i in keep], [candidate_items[i] for i in keep], {"confidence": None}

    return filt


def restore(name: str):
    """(rag, state): the restored object (ready_to_retrieve, as after prepare_retrieval_objects) and the raw record."""
    with open(os.path.join(GOLD, f"ref_state_{name}.pkl"), "rb") as f:
        st = pickle.load(f)
    rag = types.SimpleNamespace()
    rag.graph = FakeIGraph(st["vertex_names"], st["edgelist"], st["edge_weight"])
    rag.global_config = types.SimpleNamespace(**st["config"])
    rag.node_name_to_vertex_idx = dict(st["node_name_to_vertex_idx"])
    rag.passage_node_idxs = list(st["passage_node_idxs"])
    rag.passage_node_keys = list(st["passage_node_keys"])
    rag.entity_node_keys = list(st["entity_node_keys"])
    rag.fact_node_keys = list(st["fact_node_keys"])
    rag.passage_embeddings = st["passage_embeddings"]
    rag.fact_embeddings = st["fact_embeddings"]
    rag.ent_node_to_chunk_ids = {k: set(v) for k, v in st["ent_node_to_chunk_ids"].items()}
    rag.fact_embedding_store = RowStore(st["fact_rows"])
    rag.chunk_embedding_store = RowStore(st["chunk_rows"])
    rag.chunk_metadata = {}
    rag.query_to_embedding = {kind: dict(v) for kind, v in st["query_to_embedding"].items()}
    rag.rerank_filter = recorded_filter(st["queries"], st["filter_mode"])
    rag.ready_to_retrieve = True
    rag.ppr_time = rag.rerank_time = rag.all_retrieval_time = 0.0

    def get_query_embeddings(queries):          # :1391-1425 encodes what is not cached; everything asked here is
        missing = [q for q in queries if q not in rag.query_to_embedding["triple"]]
        if missing:
            raise KeyError(f"no recorded embedding for {missing}")

    def prepare_retrieval_objects():
        raise AssertionError("the restored object is already prepared")

    rag.get_query_embeddings = get_query_embeddings
    rag.prepare_retrieval_objects = prepare_retrieval_objects
    return rag, st
