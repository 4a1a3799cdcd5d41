#!/usr/bin/env python
"""Randomised soak of the mirror's batch pipeline (GPU): `HippoRAG.retrieve()` / `retrieve_dpr()` on random indices with
random max_batch, query counts (ragged last batches), num_to_retrieve (also beyond the engine's max_topk) and a
recognition-memory filter that keeps random subsets, drops everything or raises -- against the SAME calls made one batch at
a time through an engine proxy without the two-halves interface (retriever.iter_batched_retrieve then runs its serial form).
Everything must agree bit for bit: the pipeline only reorders when things are enqueued and waited for.

    python tools/soak_mirror.py [--seconds 100] [--seed 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from hipporag_amd import synth  # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float  # noqa: E402
from hipporag_amd.retriever import HippoRAG, RetrievalConfig, batched_retrieve  # noqa: E402
from tests.helpers import make_case  # noqa: E402


class Table:
    def __init__(self, t):
        self.t = t

    def batch_encode(self, texts, instruction=None, norm=True):
        return np.stack([self.t["f" if instruction and "fact" in instruction else "p"][x] for x in texts])


class SerialEngine:
    """The engine without retrieve_converged_start: iter_batched_retrieve falls back to one call at a time."""

    def __init__(self, eng):
        object.__setattr__(self, "_eng", eng)

    def __getattr__(self, name):
        if name == "retrieve_converged_start":
            raise AttributeError(name)
        return getattr(self._eng, name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=100.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    n_cases = bad = 0
    while time.time() < t_end:
        v = int(rng.choice([800, 3000, 9000]))
        e = int(v * rng.choice([4, 10]))
        seed = int(rng.integers(1, 1 << 30))
        max_batch = int(rng.choice([1, 3, 8, 17, 64, 100, 256]))
        nq = int(rng.integers(1, 4 * max_batch + 3))
        top_k = int(rng.choice([5, 40, 200]))
        prec = str(rng.choice(["bf16", "f32"]))
        mode = int(rng.integers(0, 3))                       # filter: identity / random subsets + empties / raises sometimes
        par = dict(v=v, e=e, seed=seed, max_batch=max_batch, nq=nq, top_k=top_k, prec=prec, filter=mode)
        kg, pass_bits, fact_bits, _ = make_case(v, e, 64, seed=seed)
        qf, _ = synth.make_queries_np(fact_bits, nq, seed=seed + 3)
        qp, _ = synth.make_queries_np(pass_bits, nq, seed=seed + 4)
        queries = [f"question {i}" for i in range(nq)]
        table = {"f": dict(zip(queries, bf16_bits_to_float(qf))), "p": dict(zip(queries, bf16_bits_to_float(qp)))}

        def make_filter():
            frng = {}

            def flt(query, items, indices, len_after_rerank=None):
                r = frng.setdefault(query, np.random.default_rng(abs(hash(query)) % (1 << 31) + seed % 1000))
                if mode == 0:
                    return list(indices), list(items), {}
                u = r.random()
                if mode == 2 and u < 0.1:
                    raise RuntimeError("filter failed")       # the reference swallows it into the DPR fallback (:1705-1707)
                if u < 0.25:
                    return [], [], {}
                keep = [j for j in range(len(indices)) if r.random() < 0.6]
                return [indices[j] for j in keep], [items[j] for j in keep], {}
            return flt

        pass_arg, fact_arg = pass_bits, fact_bits
        if prec == "f32":                                    # fp32 rows (not bf16-representable): the HRAG_F32_SPLIT engine
            def jitter(bits, sd):
                x = bf16_bits_to_float(bits)
                y = x + np.random.default_rng(sd).standard_normal(x.shape).astype(np.float32) * np.float32(2e-3)
                return (y / np.linalg.norm(y, axis=1, keepdims=True)).astype(np.float32)
            pass_arg, fact_arg = jitter(pass_bits, seed + 11), jitter(fact_bits, seed + 12)
        try:
            cfg = RetrievalConfig(embedding_precision=prec, max_batch=max_batch, retrieval_top_k=top_k)
            rag = HippoRAG.from_arrays(kg.csr, kg.passage_vertex, pass_arg, fact_arg, kg.subj_vertex, kg.obj_vertex,
                                       kg.num_chunks, global_config=cfg, embedding_model=Table(table), rerank_filter=make_filter())
            want_n = int(rng.choice([3, top_k, min(kg.n_passages, top_k + 37)]))
            got = rag.retrieve(queries, num_to_retrieve=want_n)
            dpr = rag.retrieve_dpr(queries, num_to_retrieve=want_n)
            # the same through the serial form of the loop, fresh filter state (same per-query random decisions)
            rag.rerank_filter = make_filter()
            rows = batched_retrieve(SerialEngine(rag.engine), queries, rag._q_tensor, rag.facts, rag.rerank_filter,
                                    linking_top_k=cfg.linking_top_k, damping=cfg.damping, passage_node_weight=cfg.passage_node_weight,
                                    ppr_iters=rag._ppr_iters(), ppr_tol=cfg.ppr_tol, ppr_max_iters=cfg.ppr_max_iters,
                                    ppr_base_iters_narrow=cfg.ppr_base_iters_narrow,
                                    num_to_retrieve=want_n, n_passages=len(rag.passage_node_keys))
            ok = len(got) == nq == len(rows) == len(dpr)
            why = "" if ok else "lengths"
            for i in range(nq):
                if not ok:
                    break
                r = rag._build_retrieval_result(queries[i], rows[i][0], rows[i][1], want_n, rows[i][2])
                if got[i].question != queries[i] or got[i].docs != r.docs or not np.array_equal(got[i].doc_scores, r.scores) \
                        or got[i].graph_seeds != r.graph_seeds:
                    ok, why = False, f"query {i} differs from the serial form"
                if len(got[i].docs) != min(want_n, kg.n_passages) or len(dpr[i].docs) != min(want_n, kg.n_passages):
                    ok, why = False, f"query {i}: {len(got[i].docs)} / {len(dpr[i].docs)} documents for {want_n}"
                if np.any(np.diff(got[i].doc_scores) > 0) or np.any(np.diff(dpr[i].doc_scores) > 0):
                    ok, why = False, f"query {i}: scores not sorted"
            rag.engine.close()
        except Exception as exc:  # noqa: BLE001
            ok, why = False, f"{type(exc).__name__}: {str(exc)[:300]}"
        n_cases += 1
        par.update(ok=ok)
        if not ok:
            bad += 1
            par["why"] = why
        print("ok  " if ok else "FAIL", json.dumps(par), flush=True)
    print(f"{n_cases} cases;", "SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
