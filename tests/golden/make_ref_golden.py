"""Golden vectors produced by the REFERENCE'S OWN CODE (tests/golden/ref_harness.py runs
/root/reference/src/hipporag with igraph / LLM / embedding model substituted, see its docstring).

    PYTHONHASHSEED=0 python tests/golden/make_ref_golden.py      (re-execs itself with the seed)

Writes tests/golden/ref_toy.npz and tests/golden/ref_synth.npz: the index arrays read off the indexed
reference object (hipporag_amd.reference_adapter.index_arrays_from_reference) and, per query, what
the reference's get_fact_scores / rerank_facts / graph_search_with_fact_entities (reset vector) /
dense_passage_retrieval / run_ppr / retrieve returned.

The reference's vertex numbering and fact order come from Python sets (extract_entity_nodes,
flatten_facts), i.e. from string hashes: PYTHONHASHSEED=0 makes the run reproducible.  The mock
embedding model emits bf16-representable fp32 vectors so that the reference (fp32 numpy) and the
device path (bf16 MFMA, fp32 accumulate) see identical inputs -- except in ref_synth_f32.npz, whose vectors are
the mock model's fp32 output as it is (the case a real embedding store presents; the engine's HRAG_F32_SPLIT mode).
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

import make_golden as mg  # noqa: E402  (toy corpus + mock embedding recipe)
import ref_harness as rh  # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float, float_to_bf16_bits  # noqa: E402
from hipporag_amd.reference_adapter import index_arrays_from_reference  # noqa: E402


class Bf16Mock(mg.MockEmbeddingModel):
    def batch_encode(self, texts, instruction=None, norm=True):
        e = super().batch_encode(texts, instruction=instruction, norm=norm)
        return bf16_bits_to_float(float_to_bf16_bits(e)).astype(np.float32)


# ----------------------------------------------------------------------------- synthetic corpus
WORDS = ["amber", "basalt", "cedar", "delta", "ember", "fjord", "garnet", "harbor", "indigo", "juniper", "kestrel",
         "lagoon", "meadow", "nectar", "onyx", "prairie", "quartz", "raven", "sierra", "tundra", "umber", "violet",
         "willow", "xenon", "yarrow", "zephyr", "atlas", "bison", "coral", "dune"]
RELS = ["founded", "borders", "supplies", "employs", "located in", "acquired", "mentors", "competes with"]


def synth_corpus(n_docs=150, n_ent=220, seed=5):
    rng = np.random.default_rng(seed)
    ents = []
    while len(ents) < n_ent:
        k = int(rng.integers(2, 4))
        name = " ".join(rng.choice(WORDS, k, replace=False)) + f" {len(ents) % 37}"
        if name not in ents:
            ents.append(name)
    # near-duplicate names -> cosine >= 0.8 under the mock model -> synonymy edges (:1006-1018)
    for i in range(0, 40, 2):
        ents.append(ents[i] + " group")
    pop = 1.0 / np.arange(1, len(ents) + 1) ** 0.8
    pop /= pop.sum()
    docs, triples = [], []
    for d in range(n_docs):
        nt = 0 if d % 29 == 7 else int(rng.integers(1, 5))        # a few passages without triples: isolated vertices
        ts = []
        for _ in range(nt):
            s, o = rng.choice(len(ents), 2, replace=False, p=pop)
            # mixed case on purpose: text_processing lower-cases (misc_utils), seeds use .lower() (:1584)
            ts.append((ents[s].title() if d % 3 == 0 else ents[s], str(rng.choice(RELS)), ents[o]))
        if d % 17 == 3 and ts:
            ts.append(ts[0])                                        # duplicate triple inside a chunk
        if d % 13 == 5 and triples and triples[-1]:
            ts.append(triples[-1][0])                               # same fact in two chunks (weight += 1, :906-910)
        text = f"Report {d}: " + "; ".join(f"{a} {r} {b}" for a, r, b in ts) if ts else f"Report {d}: nothing notable about {rng.choice(WORDS)}."
        docs.append(text)
        triples.append(ts)
    queries = []
    for qi in range(14):
        d = int(rng.integers(0, n_docs))
        while not triples[d]:
            d = int(rng.integers(0, n_docs))
        a, r, b = triples[d][0]
        queries.append(f"Which entity {r} {b.lower()}?" if qi % 2 else f"What does {a.lower()} {r}?")
    return docs, triples, queries


def make_filter(queries, mode):
    """Stands in for the LLM filter (rerank.py:108-131): it returns a SUBSET of the candidates in ITS OWN
    order.  identity / reorder+subset / drop everything (-> DPR fallback, HippoRAG.py:467-469)."""
    calls = {"cand": []}

    def filt(query, candidate_items, candidate_indices, len_after_rerank=None):
        calls["cand"].append(list(candidate_indices))
        qi = queries.index(query)
        how = "identity" if mode == "identity" else ("identity", "subset", "identity", "none", "subset")[qi % 5]
        if how == "identity":
            keep = list(range(len(candidate_indices)))
        elif how == "subset":
            keep = [i for i in (2, 0, 3) if i < len(candidate_indices)]
        else:
            keep = []
        return [candidate_indices[i] for i in keep], [candidate_items[i] for i in keep], {"confidence": None}

    return filt, calls


def dump_object_state(rag, queries, filter_mode, sols, path):
    """The indexed reference OBJECT as reference_adapter.attach() reads it (the attributes the reference's own index()
    / prepare_retrieval_objects left on it, HippoRAG.py:1287-1389), in builtin + numpy types only, and what the
    reference's own retrieve() answered: tests/restored_reference.py rebuilds the object on the GPU box, where the
    package cannot be imported, and attach() meets the real engine there."""
    import pickle
    g = rag.graph
    fact_keys = list(rag.fact_node_keys)
    state = {
        "vcount": int(g.vcount()),
        "edgelist": [(int(a), int(b)) for a, b in g.get_edgelist()],
        "edge_weight": [float(w) for w in g.es["weight"]],
        "vertex_names": [str(n) for n in g.vs["name"]],
        "node_name_to_vertex_idx": {str(k): int(v) for k, v in rag.node_name_to_vertex_idx.items()},
        "passage_node_idxs": [int(i) for i in rag.passage_node_idxs],
        "passage_node_keys": [str(k) for k in rag.passage_node_keys],
        "entity_node_keys": [str(k) for k in rag.entity_node_keys],
        "fact_node_keys": fact_keys,
        "passage_embeddings": np.asarray(rag.passage_embeddings, np.float32),
        "fact_embeddings": np.asarray(rag.fact_embeddings, np.float32),
        "ent_node_to_chunk_ids": {str(k): sorted(str(c) for c in v) for k, v in rag.ent_node_to_chunk_ids.items()},
        "fact_rows": {k: {"hash_id": r["hash_id"], "content": r["content"]}
                      for k, r in rag.fact_embedding_store.get_rows(fact_keys).items()},
        "chunk_rows": {k: {"hash_id": rag.chunk_embedding_store.get_row(k)["hash_id"],
                           "content": rag.chunk_embedding_store.get_row(k)["content"]} for k in rag.passage_node_keys},
        "config": {f: getattr(rag.global_config, f) for f in
                   ("retrieval_top_k", "linking_top_k", "damping", "passage_node_weight")},
        "query_to_embedding": {kind: {q: np.asarray(rag.query_to_embedding[kind][q], np.float32) for q in queries}
                               for kind in ("triple", "passage")},
        "queries": list(queries), "filter_mode": filter_mode,
        "reference_retrieve": [{"docs": list(s.docs), "doc_scores": np.asarray(s.doc_scores, np.float64)} for s in sols],
    }
    with open(path, "wb") as f:
        pickle.dump(state, f, protocol=4)


def run_case(name, docs, triples, queries, filter_mode, model=None, dump_state=False, save=True, **cfg):
    tmp = tempfile.mkdtemp(prefix="refgold_")
    try:
        rag = rh.build_reference_rag(tmp, docs, triples, model or Bf16Mock(), **cfg)
        filt, calls = make_filter(queries, filter_mode)
        rag.rerank_filter = filt
        sols, log = rh.capture(rag, queries)
        dpr_only = rag.retrieve_dpr(list(queries))
        a = index_arrays_from_reference(rag)
        q, np_, v = len(queries), len(a["passage_vertex"]), a["num_vertices"]
        k_f = rag.global_config.linking_top_k
        text_to_pos = {rag.chunk_embedding_store.get_row(k)["content"]: i for i, k in enumerate(rag.passage_node_keys)}
        out = {
            "num_vertices": np.int64(v), "edge_src": a["edge_src"].astype(np.int32), "edge_dst": a["edge_dst"].astype(np.int32),
            "edge_w": a["edge_w"], "passage_vertex": a["passage_vertex"], "passage_emb": a["passage_emb"],
            "fact_emb": a["fact_emb"], "subj_vertex": a["subj_vertex"], "obj_vertex": a["obj_vertex"],
            "num_chunks": a["num_chunks"],
            "qf": np.stack([rag.query_to_embedding["triple"][s] for s in queries]).astype(np.float32),
            "qp": np.stack([rag.query_to_embedding["passage"][s] for s in queries]).astype(np.float32),
            "damping": np.float64(rag.global_config.damping), "linking_top_k": np.int64(k_f),
            "passage_node_weight": np.float64(rag.global_config.passage_node_weight),
            "retrieval_top_k": np.int64(rag.global_config.retrieval_top_k),
            "fact_scores": np.stack(log["fact_scores"]).astype(np.float32),
            "dpr_ids": np.stack(log["dpr_ids"]).astype(np.int64), "dpr_scores": np.stack(log["dpr_scores"]).astype(np.float32),
        }
        cand = np.full((q, k_f), -1, np.int64); kept = np.full((q, k_f), -1, np.int64); kept_n = np.zeros(q, np.int64)
        for i in range(q):
            cand[i, :len(calls["cand"][i])] = calls["cand"][i]
            kept[i, :len(log["top_fact_idx"][i])] = log["top_fact_idx"][i]
            kept_n[i] = len(log["top_fact_idx"][i])
        used_dpr = kept_n == 0
        out.update(cand_fact_idx=cand, kept_fact_idx=kept, kept_count=kept_n, used_dpr=used_dpr)
        reset = np.zeros((q, v)); ppr_ids = np.full((q, np_), -1, np.int64); ppr_scores = np.zeros((q, np_))
        j = 0
        for i in range(q):
            if not used_dpr[i]:
                reset[i], ppr_ids[i], ppr_scores[i] = log["reset"][j], log["ppr_ids"][j], log["ppr_scores"][j]
                j += 1
        assert j == len(log["reset"])
        out.update(reset=reset, ppr_ids=ppr_ids, ppr_scores=ppr_scores)
        k_out = max(len(s.docs) for s in sols)
        final_ids = np.full((q, k_out), -1, np.int64); final_scores = np.zeros((q, k_out))
        for i, s in enumerate(sols):
            final_ids[i, :len(s.docs)] = [text_to_pos[d] for d in s.docs]
            final_scores[i, :len(s.docs)] = s.doc_scores
        out.update(final_ids=final_ids, final_scores=final_scores)
        dk = max(len(s.docs) for s in dpr_only)
        d_ids = np.full((q, dk), -1, np.int64); d_sc = np.zeros((q, dk), np.float32)
        for i, s in enumerate(dpr_only):
            d_ids[i, :len(s.docs)] = [text_to_pos[d] for d in s.docs]
            d_sc[i, :len(s.docs)] = s.doc_scores
        out.update(retrieve_dpr_ids=d_ids, retrieve_dpr_scores=d_sc)
        out["passage_texts"] = np.array([rag.chunk_embedding_store.get_row(k)["content"] for k in rag.passage_node_keys])
        out["queries"] = np.array(list(queries))
        if save:
            np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), **out)
        if dump_state:
            dump_object_state(rag, queries, filter_mode, sols, os.path.join(HERE, f"ref_state_{name}.pkl"))
        n_syn = sum(1 for w in a["edge_w"] if 0.8 <= w < 1.0)
        print(f"ref_{name}.npz: V={v} igraph edges={len(a['edge_w'])} (synonymy-like weights: {n_syn}) Np={np_} "
              f"F={len(a['subj_vertex'])} queries={q} dpr_fallbacks={int(used_dpr.sum())} "
              f"isolated passages={int(sum(1 for t in triples if not t))}")
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    assert rh.reference_available(), "needs /root/reference"
    run_case("toy", mg.DOCS, mg.TRIPLES, mg.QUERIES, "identity")
    docs, triples, queries = synth_corpus()
    run_case("synth", docs, triples, queries, "mixed")
    # the same corpus with the mock model's fp32 vectors AS THEY ARE (not bf16-representable): what a real embedding
    # store holds.  Pins the fp32-faithful similarity mode (HRAG_F32_SPLIT) -- and shows what bf16 rounding flips
    run_case("synth_f32", docs, triples, queries, "mixed", model=mg.MockEmbeddingModel(), dump_state=True)


if __name__ == "__main__":
    main()
