"""Start the MI355X engine from an existing HippoRAG working directory.

What the reference's ``index()`` leaves on disk (reference src/hipporag/):
    {save_dir}/{llm}_{embedding}/chunk_embeddings/vdb_chunk.parquet     embedding_store.py:97,160-166
    {save_dir}/{llm}_{embedding}/entity_embeddings/vdb_entity.parquet   columns: hash_id, content, embedding
    {save_dir}/{llm}_{embedding}/fact_embeddings/vdb_fact.parquet
    {save_dir}/{llm}_{embedding}/chunk_metadata.json                    HippoRAG.py:190-201
    {save_dir}/{llm}_{embedding}/graph.pickle                           igraph pickle, :225-241,1229
    {save_dir}/openie_results_ner_{llm}.json                            :178,1136-1143

``load_reference_workdir`` reads the parquet stores and the OpenIE JSON and rebuilds the graph with the
reference's own rules (add_fact_edges :867-913, add_passage_edges :915-957, add_new_nodes :1159-1187,
add_new_edges :1189-1223); synonymy edges (:959-1020) are recomputed on the GPU with
``hipporag_amd.knn`` or taken from an edge export of the pickle.  ``graph.pickle`` itself needs igraph
to read; ``tools/export_igraph_edges.py`` (run where igraph is installed) writes the ``.npz`` this
module accepts through ``graph_edges=``.
"""

from __future__ import annotations

import ast
import json
import logging
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .graph import build_csr, float_to_bf16_bits
from .retriever import HippoRAG, RetrievalConfig, compute_mdhash_id, text_processing

logger = logging.getLogger("hipporag_amd")


def load_embedding_store(directory: str, namespace: str):
    """(hash_ids, contents, embeddings fp32 [n, D]) of ``vdb_{namespace}.parquet`` in store order
    (EmbeddingStore._load_data, embedding_store.py:136-158)."""
    import pyarrow.parquet as pq
    path = os.path.join(directory, f"vdb_{namespace}.parquet")
    if not os.path.exists(path):
        return [], [], np.zeros((0, 0), np.float32)
    table = pq.read_table(path, columns=["hash_id", "content", "embedding"])
    ids = table.column("hash_id").to_pylist()
    texts = table.column("content").to_pylist()
    col = table.column("embedding").combine_chunks()
    n = len(ids)
    if n == 0:
        return ids, texts, np.zeros((0, 0), np.float32)
    flat = col.flatten() if hasattr(col, "flatten") else col.values   # list<float> -> values buffer
    emb = np.asarray(flat.to_numpy(zero_copy_only=False), dtype=np.float32)
    if emb.size % n:
        raise ValueError(f"{path}: ragged embedding column")
    return ids, texts, emb.reshape(n, emb.size // n)


def filter_invalid_triples(triples) -> List[List[str]]:
    """utils/llm_utils.py:222-254: keep 3-element triples, stringify, drop duplicates, keep order (one implementation:
    retriever.filter_invalid_triples, which index_from_openie applies to every chunk as well)."""
    from .retriever import filter_invalid_triples as _filter
    return [list(t) for t in _filter(triples)]


def load_openie_results(path: str) -> Dict[str, dict]:
    """chunk key -> {"passage", "extracted_entities", "extracted_triples"}; keys are recomputed from
    the passage text like load_existing_openie does (HippoRAG.py:1047-1054)."""
    with open(path, encoding="utf-8") as f:
        data = json.load(f)
    out = {}
    for doc in data.get("docs", []):
        out[compute_mdhash_id(doc["passage"], "chunk-")] = doc
    return out


def load_reference_workdir(save_dir: str, llm_name: str, embedding_model_name: str, *,
                           global_config: Optional[RetrievalConfig] = None, embedding_model=None,
                           rerank_filter=None, qa_fn=None, synonymy: str = "knn",
                           graph_edges: Optional[str] = None, synonymy_edge_topk: int = 2047,
                           synonymy_edge_sim_threshold: float = 0.8) -> HippoRAG:
    """A ready-to-retrieve ``hipporag_amd.HippoRAG`` from the reference's on-disk artefacts.

    synonymy: "knn" recomputes the synonymy edges on the GPU (what index() did, :959-1020),
              "none" leaves them out (CPU-only loading; the graph then lacks those edges);
    graph_edges: path of an ``.npz`` written by tools/export_igraph_edges.py (names, src, dst, weight) --
              the exact edge list of graph.pickle; overrides the rebuilt edge set when given.
    """
    llm_label = llm_name.replace("/", "_")
    work = os.path.join(save_dir, f"{llm_label}_{embedding_model_name.replace('/', '_')}")
    p_keys, p_texts, p_emb = load_embedding_store(os.path.join(work, "chunk_embeddings"), "chunk")
    e_keys, e_texts, e_emb = load_embedding_store(os.path.join(work, "entity_embeddings"), "entity")
    f_keys, f_texts, f_emb = load_embedding_store(os.path.join(work, "fact_embeddings"), "fact")
    if not p_keys:
        raise FileNotFoundError(f"no chunk store under {work}")
    openie = load_openie_results(os.path.join(save_dir, f"openie_results_ner_{llm_label}.json"))

    rag = HippoRAG(global_config, embedding_model=embedding_model, rerank_filter=rerank_filter, qa_fn=qa_fn)
    rag.passage_node_keys, rag.passage_texts = list(p_keys), list(p_texts)
    rag.entity_node_keys, rag.entity_texts = list(e_keys), list(e_texts)
    rag.fact_node_keys = list(f_keys)
    rag.facts = [tuple(ast.literal_eval(s)) for s in f_texts]            # the reference eval()s, :1693
    meta_path = os.path.join(work, "chunk_metadata.json")
    if os.path.exists(meta_path):
        with open(meta_path, encoding="utf-8") as f:
            rag.chunk_metadata = json.load(f)

    # ---- node_to_node_stats / ent_node_to_chunk_ids, exactly as index() fills them (:319-328)
    stats: Dict[Tuple[str, str], float] = {}
    ent_chunks: Dict[str, set] = {}
    for chunk_key in p_keys:
        doc = openie.get(chunk_key)
        triples = filter_invalid_triples(doc["extracted_triples"]) if doc else []      # reformat_openie_results
        triples = [text_processing(t) for t in triples]                                  # :313
        in_chunk = set()
        for t in triples:                                                                # add_fact_edges :896-913
            a, b = compute_mdhash_id(t[0], "entity-"), compute_mdhash_id(t[2], "entity-")
            in_chunk.update((a, b))
            stats[(a, b)] = stats.get((a, b), 0.0) + 1
            stats[(b, a)] = stats.get((b, a), 0.0) + 1
        for n in in_chunk:
            ent_chunks.setdefault(n, set()).add(chunk_key)
        for e in {e for t in triples for e in (t[0], t[2])}:                             # add_passage_edges :947-953
            stats[(chunk_key, compute_mdhash_id(e, "entity-"))] = 1.0
    names = list(e_keys) + list(p_keys)                                                  # add_new_nodes :1171-1175
    vid = {n: i for i, n in enumerate(names)}

    if graph_edges is not None:
        g = np.load(graph_edges, allow_pickle=False)
        g_names = [str(s) for s in g["names"]]
        remap = np.array([vid.get(n, -1) for n in g_names], np.int64)
        src, dst, wts = remap[g["src"]], remap[g["dst"]], np.asarray(g["weight"], np.float64)
        keep = (src >= 0) & (dst >= 0)
        src, dst, wts = src[keep], dst[keep], wts[keep]
    else:
        if synonymy == "knn" and len(e_keys) > 1:
            from .knn import synonymy_candidates
            for a, b, s in synonymy_candidates(e_keys, e_texts, e_emb, topk=synonymy_edge_topk,
                                               sim_threshold=synonymy_edge_sim_threshold):
                stats[(a, b)] = s                                                        # overwrites, :1013
        elif synonymy not in ("knn", "none"):
            raise ValueError(synonymy)
        src, dst, wts = [], [], []
        for (a, b), w in stats.items():                                                  # add_new_edges :1200-1223
            if a == b or a not in vid or b not in vid:
                continue
            src.append(vid[a]); dst.append(vid[b]); wts.append(w)
    rag.node_to_node_stats, rag.ent_node_to_chunk_ids = stats, ent_chunks
    rag.node_name_to_vertex_idx = vid
    csr = build_csr(len(names), src, dst, wts)

    num_chunks = np.zeros(len(names), np.int32)
    for k, s in ent_chunks.items():
        if k in vid:
            num_chunks[vid[k]] = len(s)
    ent = lambda phrase: vid.get(compute_mdhash_id(phrase.lower(), "entity-"), -1)       # :1584-1597
    subj = np.array([ent(f[0]) for f in rag.facts], np.int32)
    obj = np.array([ent(f[2]) for f in rag.facts], np.int32)
    pv = np.array([vid[k] for k in p_keys], np.int32)
    rag.entity_embeddings = e_emb if len(e_keys) else None
    # the stores' fp32 vectors as the engine takes them (RetrievalConfig.embedding_precision: fp32-faithful by default)
    rag._arrays = dict(csr=csr, passage_vertex=pv, passage_emb=rag._emb_for_engine(p_emb),
                       fact_emb=rag._emb_for_engine(f_emb) if len(f_keys) else None, subj=subj, obj=obj,
                       num_chunks=num_chunks)
    rag.ready_to_retrieve = False
    logger.info("loaded %d passages, %d entities, %d facts, %d directed entries from %s", len(p_keys), len(e_keys),
                len(f_keys), csr.nnz, work)
    return rag


def write_reference_workdir(save_dir: str, llm_name: str, embedding_model_name: str, docs: Sequence[str],
                            chunk_triples, embedding_model) -> str:
    """Write a working directory in the reference's on-disk format from documents + OpenIE triples
    (used by the tests and handy for demos; the reference writes the same files from index())."""
    import pandas as pd
    llm_label = llm_name.replace("/", "_")
    work = os.path.join(save_dir, f"{llm_label}_{embedding_model_name.replace('/', '_')}")
    m = HippoRAG(embedding_model=embedding_model).index_from_openie(list(docs), chunk_triples)

    def store(sub, ns, keys, texts):
        os.makedirs(os.path.join(work, sub), exist_ok=True)
        emb = np.asarray(embedding_model.batch_encode(list(texts)), np.float32) if len(texts) else np.zeros((0, 8))
        pd.DataFrame({"hash_id": list(keys), "content": list(texts), "embedding": [e.tolist() for e in emb]}) \
            .to_parquet(os.path.join(work, sub, f"vdb_{ns}.parquet"), index=False)

    store("chunk_embeddings", "chunk", m.passage_node_keys, m.passage_texts)
    store("entity_embeddings", "entity", m.entity_node_keys, m.entity_texts)
    store("fact_embeddings", "fact", m.fact_node_keys, [str(f) for f in m.facts])
    by_text = {}
    for d, tr in zip(docs, chunk_triples):
        by_text.setdefault(d, [list(t) for t in tr])
    docs_json = [{"idx": compute_mdhash_id(t, "chunk-"), "passage": t,
                  "extracted_entities": sorted({e for tr in by_text[t] if len(tr) == 3 for e in (tr[0], tr[2])}),
                  "extracted_triples": by_text[t]} for t in m.passage_texts]
    with open(os.path.join(save_dir, f"openie_results_ner_{llm_label}.json"), "w") as f:
        json.dump({"docs": docs_json, "avg_ent_chars": 0, "avg_ent_words": 0}, f)
    with open(os.path.join(work, "chunk_metadata.json"), "w") as f:
        json.dump(m.chunk_metadata, f)
    return work
