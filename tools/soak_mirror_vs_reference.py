#!/usr/bin/env python
"""Randomised soak of the MIRROR class's host logic against the real reference package (CPU, authoring container only:
needs /root/reference): random synthetic corpora and configurations; hipporag_amd.retriever.HippoRAG (index_from_openie,
retrieve, retrieve_dpr, retrieve_ircot) with the device engine replaced by the oracle-backed stand-in of
tests/support/adapter_on_real_reference.py, against the reference's own retrieve / retrieve_dpr / retrieve_ircot
(tests/golden/ref_harness.py; the IRCoT reasoning LLM replaced on both sides by the same deterministic function).  What
is under test is everything between the strings and the engine surface: batching, the filter loop, fallbacks, the merge
of IRCoT scores and its early stop, result materialisation.  Queries whose answer is hash-order dependent in the
reference itself (a tied link_top_k cut, HippoRAG.py:1528 / :1581) are not compared.

    python tools/soak_mirror_vs_reference.py [--cases 40] [--seed 1]"""
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
    import oracle
    from hipporag_amd import engine as engine_mod, reference_adapter as ra
    from hipporag_amd.retriever import HippoRAG, RetrievalConfig
    from tests.helpers import tie_aware_equal
    engine_mod.HippoRAGEngine = sup.OracleEngine
    rh.import_reference()
    ref_module = sys.modules["hipporag.HippoRAG"]               # the reference MODULE (the package attribute of that name is the class): its reason_step is the IRCoT LLM call

    def thought_of(query, passages, thoughts):
        """the same deterministic 'reasoning' on both sides: talks about the best passage, stops on some inputs"""
        h = sum(ord(c) for c in query) + 7 * len(thoughts)
        if h % 5 == 0 and thoughts:
            return "So the answer is: done"
        words = (passages[0] if passages else query).split()
        return " ".join(words[1 + h % 3: 7 + h % 3]) + f" regarding {query.split()[-1]}"

    rng = np.random.default_rng(args.seed)
    bad = n = 0
    for n in range(1, args.cases + 1):
        n_docs, n_ent = int(rng.integers(12, 120)), int(rng.integers(20, 200))
        cfg = dict(damping=float(rng.choice([0.5, 0.5, 0.3, 0.7])), linking_top_k=int(rng.choice([5, 5, 2, 8])),
                   passage_node_weight=float(rng.choice([0.05, 0.05, 0.01, 0.5])))
        mode = str(rng.choice(["identity", "mixed"]))
        max_batch = int(rng.choice([1, 3, 64]))
        k = int(rng.choice([3, 10, 200]))
        steps = int(rng.choice([1, 2, 3]))
        seed = int(rng.integers(1, 1 << 30))
        par = dict(n_docs=n_docs, n_ent=n_ent, seed=seed, filter=mode, max_batch=max_batch, k=k, ircot_steps=steps, **cfg)
        tmp = tempfile.mkdtemp(prefix="soak_mirror_")
        try:
            docs, triples, queries = mg.synth_corpus(n_docs, n_ent, seed)
            ref = rh.build_reference_rag(tmp, docs, triples, sup.Bf16Mock(), synonymy_edge_sim_threshold=1.5, dataset="musique", **cfg)
            cls = type(ref)
            mine = HippoRAG(RetrievalConfig(max_batch=max_batch, embedding_precision="bf16", retrieval_top_k=200, ppr_accel=False, **cfg),
                            embedding_model=sup.Bf16Mock(), rerank_filter=mg.make_filter(queries, mode)[0])
            mine.index_from_openie(docs, triples)
            # which strings are hash-order dependent in the reference: decided with the reference's own fact scores
            ref.prepare_retrieval_objects()
            arr = ra.index_arrays_from_reference(ref)
            p = oracle.column_normalize(oracle.build_symmetric_csr(arr["num_vertices"], arr["edge_src"], arr["edge_dst"], arr["edge_w"]))
            oidx = oracle.RefIndex(arr["fact_emb"], arr["passage_emb"], arr["subj_vertex"], arr["obj_vertex"], arr["num_chunks"],
                                   arr["passage_vertex"], p, linking_top_k=cfg["linking_top_k"])

            def tied(text, filt):
                fs = np.asarray(cls.get_fact_scores(ref, text))
                cand, _ = oracle.rerank_facts(fs, cfg["linking_top_k"])
                kept = [int(i) for i in filt(text, [None] * len(cand), list(cand))[0]]
                w = oracle.seed_weights(oidx, fs, kept, link_top_k=10 ** 6)[1] if kept else []
                k_l = cfg["linking_top_k"]
                return len(w) > k_l and w[k_l - 1] == w[k_l]

            def same(a, b, dense=False):
                """dense: min-max normalised cosine scores in [0, 1] -- an ABSOLUTE bar (2e-6, as tests/test_ref_golden.py):
                next to the minimum the relative error is the fp32 dot noise of the reference itself"""
                ia = [docs.index(d) for d in a.docs]
                ib = [docs.index(d) for d in b.docs]
                gap = dict(rel_gap=0.0, abs_gap=4e-6) if dense else dict(rel_gap=2e-5)
                if len(ia) != len(ib) or not tie_aware_equal(ia, ib, np.asarray(b.doc_scores), **gap):
                    return "documents differ"
                tol = dict(rtol=0, atol=2e-6) if dense else dict(rtol=2e-5, atol=1e-7)   # atol: the fp32 dot noise in the prior of a near-minimum passage (tests/helpers.prior_noise_allowance)
                if not np.allclose(np.sort(a.doc_scores)[::-1], np.sort(b.doc_scores)[::-1], **tol):
                    return "scores differ"
                return ""

            why = ""
            ref.rerank_filter = mg.make_filter(queries, mode)[0]
            r_ret = ref.retrieve(list(queries), num_to_retrieve=k)
            m_ret = mine.retrieve(list(queries), num_to_retrieve=k)
            probe = mg.make_filter(queries, mode)[0]
            skipped = 0
            for qi, q in enumerate(queries):
                if tied(q, probe):
                    skipped += 1
                    continue
                w = same(m_ret[qi], r_ret[qi])
                if w or [tuple(x) for x in m_ret[qi].graph_seeds] != [tuple(x) for x in r_ret[qi].graph_seeds]:
                    why = f"retrieve, query {qi}: {w or 'graph_seeds differ'}"
            r_dpr, m_dpr = ref.retrieve_dpr(list(queries), num_to_retrieve=k), mine.retrieve_dpr(list(queries), num_to_retrieve=k)
            for qi in range(len(queries)):
                w = same(m_dpr[qi], r_dpr[qi], dense=True)
                if w:
                    why = f"retrieve_dpr, query {qi}: {w}"
            # IRCoT: the filter of make_filter is keyed on the ORIGINAL queries; thoughts are new strings -> identity filter
            ident = lambda query, items, idx, len_after_rerank=None: (list(idx), list(items), {})
            ref.rerank_filter, mine.rerank_filter = ident, ident
            used = []
            ref_module.reason_step = lambda dataset, ptm, query, passages, thoughts, llm: (used.append(query), thought_of(query, passages, thoughts))[1]
            r_ir = ref.retrieve_ircot(list(queries), max_qa_steps=steps, num_to_retrieve=k)
            m_ir = mine.retrieve_ircot(list(queries), max_qa_steps=steps, num_to_retrieve=k, reason_fn=thought_of)
            n_ir = 0
            for qi, q in enumerate(queries):
                strings = [q] + list(r_ir[qi].thoughts)
                if any(tied(t, ident) for t in strings if "So the answer is:" not in t) or list(m_ir[qi].thoughts) != list(r_ir[qi].thoughts):
                    # a tied constituent retrieval (or thoughts that already diverged through one): not comparable
                    if list(m_ir[qi].thoughts) != list(r_ir[qi].thoughts) and not any(tied(t, ident) for t in [q] + list(m_ir[qi].thoughts) + list(r_ir[qi].thoughts) if "So the answer is:" not in t):
                        why = f"retrieve_ircot, query {qi}: thoughts differ without a tie"
                    continue
                n_ir += 1
                w = same(m_ir[qi], r_ir[qi])
                if w:
                    why = f"retrieve_ircot, query {qi}: {w}"
            par.update(ok=not why, hash_order_dependent=skipped, ircot_compared=n_ir)
            if why:
                par["why"] = why
        except Exception as exc:  # noqa: BLE001
            par.update(ok=False, error=f"{type(exc).__name__}: {str(exc)[:300]}", trace=traceback.format_exc()[-900:])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        bad += 0 if par["ok"] else 1
        print("ok  " if par["ok"] else "FAIL", json.dumps(par), flush=True)
    print(f"{n} cases;", "SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
