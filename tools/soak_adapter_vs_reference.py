#!/usr/bin/env python
"""Randomised soak of reference_adapter.attach() on the REAL reference object (CPU, authoring container only: needs
/root/reference): random synthetic corpora, configurations and filter behaviours (tests/golden/make_ref_golden.py); the
reference's own retrieve() before attach() against the adapter's batched route after it, with the device engine replaced
by the oracle-backed stand-in of tests/support/adapter_on_real_reference.py -- what is under test is the host logic
between the reference class and the engine surface (batches, the filter loop, fallbacks, result classes, seams, detach).

    python tools/soak_adapter_vs_reference.py [--cases 40] [--seed 1]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import ref_harness as rh
    if not rh.reference_available():
        print("the reference sources are not present: nothing to do")
        return 0
    import adapter_on_real_reference as sup
    import make_ref_golden as mg
    from hipporag_amd import engine as engine_mod, reference_adapter as ra
    import oracle
    from tests.helpers import tie_aware_equal
    engine_mod.HippoRAGEngine = sup.OracleEngine
    rng = np.random.default_rng(args.seed)
    bad = n = 0
    for n in range(1, args.cases + 1):
        n_docs, n_ent = int(rng.integers(12, 160)), int(rng.integers(20, 260))
        cfg = dict(damping=float(rng.choice([0.5, 0.5, 0.3, 0.7])), linking_top_k=int(rng.choice([5, 5, 2, 8])),
                   passage_node_weight=float(rng.choice([0.05, 0.05, 0.01, 0.5])))
        mode = str(rng.choice(["identity", "mixed"]))
        max_batch = int(rng.choice([1, 2, 5, 64]))
        k = int(rng.choice([3, 10, 200]))
        seed = int(rng.integers(1, 1 << 30))
        par = dict(n_docs=n_docs, n_ent=n_ent, seed=seed, filter=mode, max_batch=max_batch, k=k, **cfg)
        tmp = tempfile.mkdtemp(prefix="soak_adapter_")
        t0 = time.time()
        try:
            docs, triples, queries = mg.synth_corpus(n_docs, n_ent, seed)
            rag = rh.build_reference_rag(tmp, docs, triples, sup.Bf16Mock(), **cfg)
            rag.rerank_filter = mg.make_filter(queries, mode)[0]
            before = rag.retrieve(list(queries), num_to_retrieve=k)
            dpr_before = rag.retrieve_dpr(list(queries), num_to_retrieve=k)
            cls = type(rag)
            ra.attach(rag, max_batch=max_batch)
            rag.rerank_filter = mg.make_filter(queries, mode)[0]
            after = rag.retrieve(list(queries), num_to_retrieve=k)
            pos = {rag.chunk_embedding_store.get_row(key)["content"]: i for i, key in enumerate(rag.passage_node_keys)}
            why = ""
            # where the reference's own answer is hash-order dependent: the link_top_k cut of get_top_k_weights inside a run
            # of equal phrase weights (HippoRAG.py:1528, :1581; tests/test_ref_golden.py skips those queries too)
            arr = ra.index_arrays_from_reference(rag)
            p = oracle.column_normalize(oracle.build_symmetric_csr(arr["num_vertices"], arr["edge_src"], arr["edge_dst"], arr["edge_w"]))
            oidx = oracle.RefIndex(arr["fact_emb"], arr["passage_emb"], arr["subj_vertex"], arr["obj_vertex"], arr["num_chunks"],
                                   arr["passage_vertex"], p, linking_top_k=cfg["linking_top_k"])
            filt = mg.make_filter(queries, mode)[0]
            tied = []
            for q in queries:
                fs = np.asarray(cls.get_fact_scores(rag, q))
                cand, _ = oracle.rerank_facts(fs, cfg["linking_top_k"])
                kept = [int(i) for i in filt(q, [None] * len(cand), list(cand))[0]]
                ids_w = oracle.seed_weights(oidx, fs, kept, link_top_k=10 ** 6)[1] if kept else []
                k_l = cfg["linking_top_k"]
                tied.append(len(ids_w) > k_l and ids_w[k_l - 1] == ids_w[k_l])
            par["hash_order_dependent_queries"] = int(sum(tied))
            for qi, (a, b) in enumerate(zip(after, before)):
                if tied[qi]:
                    continue
                ia, ib = [pos[d] for d in a.docs], [pos[d] for d in b.docs]
                if len(ia) != len(ib) or not tie_aware_equal(ia, ib, np.asarray(b.doc_scores), rel_gap=2e-5):
                    why = f"query {qi}: documents differ"
                elif not np.allclose(np.sort(a.doc_scores)[::-1], np.sort(b.doc_scores)[::-1], rtol=2e-5, atol=1e-7):     # atol: the fp32 dot noise in the prior of near-minimum passages (tests/helpers.prior_noise_allowance)
                    why = f"query {qi}: scores differ"
                elif a.question != b.question or [tuple(x) for x in a.graph_seeds] != [tuple(x) for x in b.graph_seeds]:
                    why = f"query {qi}: question / graph_seeds differ"
            q0 = queries[0]                          # the per-method seams against the reference's own methods
            if not np.allclose(rag.get_fact_scores(q0), cls.get_fact_scores(rag, q0), atol=2e-6, rtol=0):
                why = "get_fact_scores seam"
            if not np.allclose(rag.dense_passage_retrieval(q0)[1], cls.dense_passage_retrieval(rag, q0)[1], atol=2e-6, rtol=0):
                why = "dense_passage_retrieval seam"
            ra.detach(rag)
            # the other route: only the per-method seams replaced, the reference's own retrieve() loop drives them
            ra.attach(rag, max_batch=max_batch, batched_retrieve=False)
            rag.rerank_filter = mg.make_filter(queries, mode)[0]
            seams = rag.retrieve(list(queries), num_to_retrieve=k)
            for qi, (a, b) in enumerate(zip(seams, before)):
                if tied[qi]:
                    continue
                ia, ib = [pos[d] for d in a.docs], [pos[d] for d in b.docs]
                if len(ia) != len(ib) or not tie_aware_equal(ia, ib, np.asarray(b.doc_scores), rel_gap=2e-5) or \
                        not np.allclose(np.sort(a.doc_scores)[::-1], np.sort(b.doc_scores)[::-1], rtol=2e-5, atol=1e-7):
                    why = f"seams route, query {qi}: differs from the reference"
            ra.detach(rag)
            again = rag.retrieve_dpr(list(queries), num_to_retrieve=k)
            if [s.docs for s in again] != [s.docs for s in dpr_before]:
                why = "after detach the reference's own methods do not answer as before"
            par.update(ok=not why, seconds=round(time.time() - t0, 1))
            if why:
                par["why"] = why
        except Exception as exc:  # noqa: BLE001
            par.update(ok=False, error=f"{type(exc).__name__}: {str(exc)[:300]}", trace=traceback.format_exc()[-700:])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        bad += 0 if par["ok"] else 1
        print("ok  " if par["ok"] else "FAIL", json.dumps(par), flush=True)
    print(f"{n} cases;", "SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
