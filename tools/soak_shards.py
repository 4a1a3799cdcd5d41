#!/usr/bin/env python
"""Randomised differential soak of the multi-GPU modes on ONE device (GPU): random indices x world sizes x batch sizes x
exchange groups x damping x (fixed count / convergence contract), all ranks as threads (dist.run_local_shards /
dist.run_local_hybrid: the code every rank of a real N-GPU job runs, with LocalComm standing in for RCCL), checked
against the oracle on a few queries per case and -- hybrid mode -- bit for bit against the single-GPU engine.

    python tools/soak_shards.py [--seconds 120] [--seed 1]"""
import argparse
import dataclasses
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import oracle  # noqa: E402
from hipporag_amd import dist as hd, synth  # noqa: E402
from hipporag_amd.engine import HippoRAGEngine  # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float  # noqa: E402
from hipporag_amd.retriever import sweeps_for_damping  # noqa: E402
from tests.helpers import make_case, prior_noise_allowance, ranked_parity  # noqa: E402


def bf16(bits, dev):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(dev).view(torch.bfloat16)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: only the time bound): a seeded run of a fixed\n                    number of cases is the same on every box")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak_shards.json"))
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda", 0)
    t_end = time.time() + args.seconds
    cases, bad = [], 0
    while time.time() < t_end and not (args.cases and len(cases) >= args.cases):
        v = int(rng.choice([3000, 8000, 16000]))
        e = int(v * rng.choice([3, 8, 15]))
        world = int(rng.choice([2, 3, 4, 8]))
        mode = str(rng.choice(["rowshard", "rowshard", "hybrid"]))
        native = bool(rng.integers(0, 2)) if mode == "rowshard" else False   # the host loop inside the library (round 6)
        b = int(rng.choice([40, 66, 128, 130, 256, 300]))
        if mode == "hybrid":
            b = max(world, b // world * world)            # the hybrid deals the batch evenly
        groups = int(rng.choice([0, 1, 2]))
        damping = float(rng.choice([0.3, 0.5, 0.5, 0.6]))
        iters = sweeps_for_damping(damping)
        tol = float(rng.choice([0.0, 1.5e-6])) if mode == "rowshard" else 0.0
        power_law = bool(rng.integers(0, 2))
        seed = int(rng.integers(1, 1 << 30))
        par = dict(v=v, e=e, world=world, mode=mode, b=b, groups=groups, damping=damping, iters=iters, tol=tol,
                   power_law=power_law, seed=seed, native=native)
        kg, pass_bits, fact_bits, index = make_case(v, e, 64, seed=seed, power_law=power_law)
        index = dataclasses.replace(index, damping=damping)
        qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=seed + 3)
        qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=seed + 4)
        qf_t, qp_t = bf16(qf_bits, dev), bf16(qp_bits, dev)
        k_docs = min(100, kg.n_passages)
        kw = dict(link_top_k=5, damping=damping, passage_node_weight=0.05, ppr_iters=iters, k=k_docs)
        if tol > 0:
            kw.update(ppr_tol=tol, ppr_max_iters=29)
        try:
            sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
            if mode == "rowshard":
                got = hd.run_local_shards(world, sidx, pass_bits, fact_bits, qf_t, qp_t, kw, groups, dev, k_docs, native=native)
                d_idx, d_sc, flags = got[2], got[3], got[4]
                same_as_single = None
                if native and rng.random() < 0.25:       # and bit for bit the Python host loop, on a quarter of the cases
                    ref_loop = hd.run_local_shards(world, sidx, pass_bits, fact_bits, qf_t, qp_t, kw, groups, dev, k_docs)
                    if not all(np.array_equal(a, b_) for a, b_ in zip(got, ref_loop)):
                        raise AssertionError("hrag_shard_retrieve differs from the Python host loop")
            else:
                arrays = dict(csr=kg.csr, passage_vertex=kg.passage_vertex, subj_vertex=kg.subj_vertex,
                              obj_vertex=kg.obj_vertex, num_chunks=kg.num_chunks)
                got = hd.run_local_hybrid(world, arrays, sidx, pass_bits, fact_bits, qf_t, qp_t, kw, dev, k_docs)
                d_idx, d_sc, flags = got[2], got[3], got[4]
                with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                                    kg.num_chunks, max_batch=b // world, max_topk=k_docs) as one:
                    same_as_single = True
                    bn = b // world
                    for r in range(world):             # every rank's share through the single-GPU engine: bit for bit
                        rows = slice(r * bn, (r + 1) * bn)
                        i1, s1 = one.score_facts(qf_t[rows], k=5)
                        o1 = one.retrieve(qp_t[rows], i1, s1, torch.full((bn,), 5, dtype=torch.int32, device=dev), **kw)
                        torch.cuda.synchronize()
                        same_as_single &= bool(np.array_equal(o1.doc_idx.cpu().numpy(), d_idx[rows]) and
                                               np.array_equal(o1.doc_score.cpu().numpy(), d_sc[rows]) and
                                               np.array_equal(i1.cpu().numpy(), got[0][rows]))
        except Exception as exc:  # noqa: BLE001
            bad += 1
            par["error"] = f"{type(exc).__name__}: {str(exc)[:300]}"
            print("FAIL", json.dumps(par), flush=True)
            cases.append(par)
            continue
        ok, why, worst = True, "", 0.0
        if same_as_single is False:
            ok, why = False, "hybrid result differs from the single-GPU engine"
        bad_flags = flags & ~(16 if tol > 0 else 0)       # a slow graph may leave NOT_CONVERGED for the host repeat
        if np.any(bad_flags):
            ok, why = False, f"flags {sorted(set(int(f) for f in flags))}"
        qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
        for q in sorted(set(np.linspace(0, b - 1, 4).astype(int).tolist())):
            if flags[q] & 16:
                continue
            ref = oracle.retrieve_one(index, qf[q], qp[q], ppr_mode="exact")
            full = ref.x[kg.passage_vertex]
            rep = ranked_parity(d_idx[q], d_sc[q], ref.sorted_doc_ids, ref.sorted_doc_scores, full)
            if tol == 0.0 and rep["worst_rel_err"] >= 1e-5:      # a fixed count is only as accurate as the graph mixes
                ref = oracle.retrieve_one(index, qf[q], qp[q], ppr_mode="power", ppr_iters=iters)
                full = ref.x[kg.passage_vertex]
                rep = ranked_parity(d_idx[q], d_sc[q], ref.sorted_doc_ids, ref.sorted_doc_scores, full)
            allow = float(prior_noise_allowance(index, qp[q])[d_idx[q]].max())
            worst = max(worst, rep["worst_rel_err"])
            if rep["worst_rel_err"] >= 1e-5 + allow or (not rep["equal"] and allow < 1e-6):
                ok, why = False, f"q{q}: err {rep['worst_rel_err']:.2e} (allow {allow:.1e}) equal={rep['equal']}"
        par.update(ok=ok, worst=worst, bit_identical_to_single=same_as_single)
        if not ok:
            bad += 1
            par["why"] = why
        print("ok  " if ok else "FAIL", json.dumps(par), flush=True)
        cases.append(par)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"cases": cases, "failed": bad}, open(args.out, "w"), indent=0)
    print(f"{len(cases)} cases;", "SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
