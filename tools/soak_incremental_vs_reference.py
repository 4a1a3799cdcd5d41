#!/usr/bin/env python
"""Randomised soak of the mirror's INCREMENTAL index / delete (graph compiler) against the real reference
package (CPU, authoring container only: needs /root/reference): random synthetic corpora (tests/golden/make_ref_golden.py),
random life cycles  index(A) -> index(B, overlapping A) -> delete(some) [-> index(C, re-adding deleted documents)], run by
the reference's own code (tests/golden/ref_harness.py) and by hipporag_amd.retriever.HippoRAG.index_from_openie / delete;
after EVERY step -- and, for the fresh index, once more for the reference's working directory read back by hipporag_amd.loaders -- the two must hold the same thing BY NAME: vertex set, the igraph edge list with parallel edges summed,
passage store order and texts, fact store contents, the chunk-count divisor of every entity (the assertions of
tests/test_incremental_index.py::test_mirror_life_cycle_matches_the_reference_by_name).

    python tools/soak_incremental_vs_reference.py [--cases 40] [--seed 1]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def summed(names, src, dst, w):
    out = {}
    for a, b, x in zip(src, dst, w):
        k = tuple(sorted((names[a], names[b]))) if names is not None else tuple(sorted((a, b)))
        out[k] = out.get(k, 0.0) + float(x)
    return out


def compare(rag_ref, rag):
    rag_ref.prepare_retrieval_objects()
    g = rag_ref.graph
    names = [v["name"] for v in g.vs]
    es = g.get_edgelist()
    w = list(g.es["weight"]) if es else []
    mg_ = rag._graph
    if set(mg_.names) != set(names):
        return "vertex sets differ"
    s, d, ww = mg_.edge_list()
    mine = summed(mg_.names, s.tolist(), d.tolist(), ww.tolist())
    ref = summed(None, [names[a] for a, _ in es], [names[b] for _, b in es], w)
    if set(mine) != set(ref) or any(abs(mine[k] - ref[k]) > 2e-6 * max(1.0, abs(ref[k])) for k in ref):
        return "edge lists differ"
    if rag.passage_node_keys != list(rag_ref.passage_node_keys):
        return "passage store order differs"
    if rag.passage_texts != [rag_ref.chunk_embedding_store.get_row(k)["content"] for k in rag_ref.passage_node_keys]:
        return "passage texts differ"
    rows = rag_ref.fact_embedding_store.get_rows(list(rag_ref.fact_node_keys)) if len(rag_ref.fact_node_keys) else {}
    if {str(f) for f in rag.facts} != {rows[k]["content"] for k in rag_ref.fact_node_keys}:
        return "fact stores differ"
    if {k: len(v) for k, v in rag.ent_node_to_chunk_ids.items()} != {k: len(v) for k, v in rag_ref.ent_node_to_chunk_ids.items()}:
        return "chunk counts differ"
    a = rag._arrays
    if a["csr"].num_vertices != len(names) or a["passage_emb"].shape[0] != len(rag.passage_node_keys) or \
            a["fact_emb"].shape[0] != len(rag.facts):
        return "array shapes differ"
    return ""


def synonymy_edges_cpu(keys, texts, embs, *, topk=2047, sim_threshold=0.8):
    """The contract of hipporag_amd.knn.synonymy_candidates (add_synonymy_edges, HippoRAG.py:959-1020) with a numpy KNN in the
    place of the GPU one: normalised fp32 vectors, cosine, neighbours in falling score order down to the threshold, at most
    101 per entity, not itself, entities with <= 2 alphanumerics are skipped as queries.  Returns [(key a, key b, score)] --
    what HippoRAG.index_from_openie(synonymy=<callable>) consumes."""
    import re
    texts = list(texts)
    if not texts:
        return []
    e = np.asarray(embs, np.float32)
    e = e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12)
    s = (e @ e.T).astype(np.float32)
    out = []
    for i, t in enumerate(texts):
        if len(re.sub("[^A-Za-z0-9]", "", t)) <= 2:
            continue
        order = np.argsort(-s[i], kind="stable")[:topk]
        n = 0
        for j in order:
            if s[i, j] < sim_threshold or n > 100:
                break
            if j != i and texts[j] != "":
                out.append((keys[i], keys[j], float(s[i, j])))
                n += 1
    return out


def same_index(a, b):
    """Two mirror objects (one indexed in memory and equal to the reference by name, one loaded from the reference's
    working directory) hold the same index up to the vertex numbering: matrix entries by name, stores, divisors, rows."""
    def by_name(rag):
        csr, inv = rag._arrays["csr"], {v: k for k, v in rag.node_name_to_vertex_idx.items()}
        rows = np.repeat(np.arange(csr.num_vertices), np.diff(csr.row_ptr))
        return {(inv[int(r)], inv[int(c)]): float(w) for r, c, w in zip(rows, csr.col_idx, csr.raw)}
    ea, eb = by_name(a), by_name(b)
    if set(ea) != set(eb) or any(abs(ea[k] - eb[k]) > 1e-9 * max(1.0, abs(ea[k])) for k in ea):
        return "matrix entries differ by name"
    if a.passage_node_keys != b.passage_node_keys or a.passage_texts != b.passage_texts:
        return "passage stores differ"
    if set(a.facts) != set(b.facts) or set(a.entity_node_keys) != set(b.entity_node_keys):
        return "fact / entity stores differ"
    nc = lambda r: {k: int(r._arrays["num_chunks"][v]) for k, v in r.node_name_to_vertex_idx.items() if k.startswith("entity-")}
    if nc(a) != nc(b):
        return "chunk-count divisors differ"
    if not np.array_equal(a._arrays["passage_emb"], b._arrays["passage_emb"]):
        return "passage embedding rows differ"
    fa = {f: a._arrays["fact_emb"][i].tobytes() for i, f in enumerate(a.facts)}
    fb = {f: b._arrays["fact_emb"][i].tobytes() for i, f in enumerate(b.facts)}
    if fa != fb:
        return "fact embedding rows differ"
    return ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import ref_harness as rh
    if not rh.reference_available():
        print("the reference sources are not present: nothing to do")
        return 0
    import make_ref_golden as mgold
    from make_ref_incremental import Bf16Mock
    from hipporag_amd.retriever import HippoRAG, RetrievalConfig
    rng = np.random.default_rng(args.seed)
    bad = n = 0
    for n in range(1, args.cases + 1):
        n_docs, n_ent = int(rng.integers(8, 90)), int(rng.integers(20, 120))
        seed = int(rng.integers(1, 1 << 30))
        docs, triples, _ = mgold.synth_corpus(n_docs, n_ent, seed)
        # synth_corpus can produce the same text twice only by accident; keep texts unique (the stores hash them)
        seen, keep = set(), []
        for i, d in enumerate(docs):
            if d not in seen:
                seen.add(d)
                keep.append(i)
        docs, triples = [docs[i] for i in keep], [triples[i] for i in keep]
        n_docs = len(docs)
        cut = int(rng.integers(1, n_docs))
        step_a = list(range(0, cut))
        step_b = list(range(int(rng.integers(0, cut)), n_docs))                  # overlaps A: those must collapse
        delete = sorted(rng.choice(n_docs, int(rng.integers(1, max(2, n_docs // 3))), replace=False).tolist())
        readd = bool(rng.integers(0, 2))
        syn = bool(rng.integers(0, 2))                       # synonymy edges on: the mirror takes them as an explicit list
        thr = 0.8 if syn else 1.5
        par = dict(n_docs=n_docs, n_ent=n_ent, seed=seed, cut=cut, b_from=step_b[0], n_delete=len(delete), readd=readd, synonymy=syn)
        tmp = tempfile.mkdtemp(prefix="soak_inc_")
        try:
            ref = rh.build_reference_rag(tmp, [docs[i] for i in step_a], [triples[i] for i in step_a], Bf16Mock(),
                                         synonymy_edge_sim_threshold=thr)
            mine = HippoRAG(RetrievalConfig(max_batch=4, embedding_precision="bf16"), embedding_model=Bf16Mock())

            def index_mine(ids):
                """index_from_openie; with synonymy on, like the reference: the KNN over ALL entities the store holds after the
                call decides the synonymy edges of the call (synonymy=<callable>: the numpy stand-in of the GPU KNN)"""
                mine.index_from_openie([docs[i] for i in ids], [triples[i] for i in ids],
                                       synonymy=synonymy_edges_cpu if syn else None, synonymy_edge_sim_threshold=thr)

            index_mine(step_a)
            why = compare(ref, mine)
            step = "a"
            if not why and not syn:
                # the reference's working directory of this FRESH index (parquet stores, OpenIE json, chunk metadata) through
                # the on-disk loader, which rebuilds the graph from them: the same index by name.  (After an
                # incremental life cycle the reference's graph carries history the files do not -- surviving edges are not
                # decremented by delete() -- and the loader then takes the exported edge list of graph.pickle instead.)
                from hipporag_amd.loaders import load_reference_workdir
                loaded = load_reference_workdir(tmp, ref.global_config.llm_name, ref.global_config.embedding_model_name,
                                                synonymy="none", embedding_model=Bf16Mock(),
                                                global_config=RetrievalConfig(max_batch=4, embedding_precision="bf16"))
                why, step = same_index(mine, loaded), "a (loader)"
            if not why:
                ref.global_config.force_openie_from_scratch = False
                ref.global_config.force_index_from_scratch = False
                ref.openie = rh.FixedOpenIE({d: t for d, t in zip(docs, triples)})
                ref.index([docs[i] for i in step_b])
                index_mine(step_b)
                why, step = compare(ref, mine), "b"
            if not why:
                ref.delete([docs[i] for i in delete])
                mine.delete([docs[i] for i in delete])
                why, step = compare(ref, mine), "c"
            if not why and readd:
                again = delete[: max(1, len(delete) // 2)]
                ref.index([docs[i] for i in again])
                index_mine(again)
                why, step = compare(ref, mine), "d"
            par.update(ok=not why, V=len(mine._graph.names), facts=len(mine.facts))
            if why:
                par["why"] = f"after step {step}: {why}"
        except Exception as exc:  # noqa: BLE001
            par.update(ok=False, error=f"{type(exc).__name__}: {str(exc)[:300]}", trace=traceback.format_exc()[-800:])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        bad += 0 if par["ok"] else 1
        print("ok  " if par["ok"] else "FAIL", json.dumps(par), flush=True)
    print(f"{n} cases;", "SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
