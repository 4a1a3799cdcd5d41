"""A stand-in for an indexed reference ``hipporag.HippoRAG`` object (the real package cannot be
imported here: igraph / openai / ... are absent).  It carries exactly the attributes the reference's
``prepare_retrieval_objects`` (src/hipporag/HippoRAG.py:1287-1389) leaves on the object, built with
the reference's rules from documents + OpenIE triples:
  * one igraph edge per ``node_to_node_stats`` key (:1189-1223) -- so a fact pair (s,o)/(o,s) is TWO
    parallel undirected edges, which the CSR builder has to sum;
  * vertices = entity keys then passage keys (:1171-1175);
  * stores return rows ``{"hash_id", "content"}`` (embedding_store.py).
"""

from __future__ import annotations

import types

import numpy as np

from hipporag_amd.retriever import HippoRAG as Mirror, identity_rerank_filter


class FakeIGraph:
    def __init__(self, names, edges, weights):
        self._names, self._edges = list(names), list(edges)
        self.es = {"weight": list(weights)}
        self.vs = [{"name": n} for n in names]

    def vcount(self):
        return len(self._names)

    def get_edgelist(self):
        return list(self._edges)


class FakeStore:
    def __init__(self, keys, contents, embeddings=None):
        self._rows = {k: {"hash_id": k, "content": c} for k, c in zip(keys, contents)}
        self._emb = None if embeddings is None else {k: e for k, e in zip(keys, embeddings)}

    def get_all_ids(self):
        return list(self._rows)

    def get_row(self, key):
        return self._rows[key]

    def get_rows(self, keys):
        return {k: self._rows[k] for k in keys}

    def get_embeddings(self, keys):
        return [self._emb[k] for k in keys]


def make_fake_reference(docs, triples, embedding_model, *, linking_top_k=5, retrieval_top_k=200):
    m = Mirror(embedding_model=embedding_model).index_from_openie(docs, triples)
    names = m.entity_node_keys + m.passage_node_keys
    vid = {n: i for i, n in enumerate(names)}
    edges, weights = [], []
    for (a, b), w in m.node_to_node_stats.items():           # add_new_edges :1200-1223
        if a == b or a not in vid or b not in vid:
            continue
        edges.append((vid[a], vid[b]))
        weights.append(float(w))
    pe = np.asarray(embedding_model.batch_encode(m.passage_texts), np.float32)
    fe = np.asarray(embedding_model.batch_encode([str(f) for f in m.facts]), np.float32)
    rag = types.SimpleNamespace()
    rag.graph = FakeIGraph(names, edges, weights)
    rag.global_config = types.SimpleNamespace(linking_top_k=linking_top_k, retrieval_top_k=retrieval_top_k,
                                              damping=0.5, passage_node_weight=0.05)
    rag.chunk_embedding_store = FakeStore(m.passage_node_keys, m.passage_texts, pe)
    rag.fact_embedding_store = FakeStore(m.fact_node_keys, [str(f) for f in m.facts], fe)
    rag.chunk_metadata = {}
    rag.rerank_filter = identity_rerank_filter
    rag.embedding_model = embedding_model
    rag.ready_to_retrieve = False
    rag.ppr_time = rag.rerank_time = rag.all_retrieval_time = 0.0

    def prepare_retrieval_objects():                          # :1287-1389, the parts the path reads
        rag.query_to_embedding = {"triple": {}, "passage": {}}
        rag.passage_node_keys = rag.chunk_embedding_store.get_all_ids()
        rag.fact_node_keys = rag.fact_embedding_store.get_all_ids()
        rag.node_name_to_vertex_idx = {node["name"]: i for i, node in enumerate(rag.graph.vs)}
        rag.passage_node_idxs = [rag.node_name_to_vertex_idx[k] for k in rag.passage_node_keys]
        rag.passage_embeddings = np.array(rag.chunk_embedding_store.get_embeddings(rag.passage_node_keys))
        rag.fact_embeddings = np.array(rag.fact_embedding_store.get_embeddings(rag.fact_node_keys))
        rag.ent_node_to_chunk_ids = dict(m.ent_node_to_chunk_ids)
        rag.ready_to_retrieve = True

    def get_query_embeddings(queries):                        # :1391-1425
        new = [q for q in queries if q not in rag.query_to_embedding["triple"]]
        if new:
            for kind, instr in (("triple", "query_to_fact"), ("passage", "query_to_passage")):
                for q, e in zip(new, embedding_model.batch_encode(new, instruction=instr, norm=True)):
                    rag.query_to_embedding[kind][q] = np.asarray(e, np.float32)

    rag.prepare_retrieval_objects = prepare_retrieval_objects
    rag.get_query_embeddings = get_query_embeddings
    return rag
