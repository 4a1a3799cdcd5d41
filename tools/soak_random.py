#!/usr/bin/env python
"""Randomised differential soak (GPU): random small indices x batch sizes x damping x sweep plans (plain / accelerated,
fixed count / convergence contract) x filter outcomes (all kept, subsets, nothing kept -> DPR ranking) through
hrag_score_facts + hrag_retrieve, every case checked against the oracle on a few queries (ranked ids tie-class aware,
scores <= 1e-5 relative -- plus the prior-noise allowance of tests/helpers.py where the reference itself is undefined).

    python tools/soak_random.py [--seconds 150] [--seed 1]

Prints one line per case and `SOAK OK` / `SOAK FAILED (n)`; the failures' parameters are enough to replay them."""
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
from hipporag_amd import _lib, synth  # noqa: E402
from hipporag_amd.engine import HippoRAGEngine  # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float  # noqa: E402
from hipporag_amd.retriever import sweeps_for_damping  # noqa: E402
from tests.helpers import make_case, prior_noise_allowance, ranked_parity, tie_aware_equal  # noqa: E402


def bf16(bits, dev):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(dev).view(torch.bfloat16)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: only the time bound): a seeded run of a fixed\n                    number of cases is the same on every box")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak_random.json"))
    ap.add_argument("--replay", action="append", default=[], help="a case as printed (JSON): run exactly that one; repeatable")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--big", action="store_true", help="50k .. 200k vertices, batches up to 512, up to 2048 documents per query")
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda", 0)
    t_end = time.time() + args.seconds
    cases, bad = [], 0
    replay = [json.loads(x) for x in args.replay]
    while time.time() < t_end and (not args.replay or replay) and not (args.cases and len(cases) >= args.cases):
        forced = replay.pop(0) if args.replay else None
        if forced is not None:                      # the case's own stream of random decisions restarts from its seed
            rng = np.random.default_rng(forced["seed"])
        v = int(rng.choice([50000, 100000, 200000] if args.big else [600, 1500, 4000, 9000, 20000]))
        e = int(v * rng.choice([3, 6, 10, 20]))
        dim = int(rng.choice([32, 64, 96, 192]))
        power_law = bool(rng.integers(0, 2))
        pfrac = float(rng.choice([0.05, 0.125, 0.3]))
        b = int(rng.choice([1, 8, 33, 64, 130, 256, 384, 512] if args.big else [1, 2, 3, 5, 8, 9, 17, 40, 64, 65, 100, 128, 130, 256, 300]))
        damping = float(rng.choice([0.2, 0.3, 0.4, 0.45, 0.5, 0.5, 0.5, 0.55, 0.6, 0.7]))
        iters = sweeps_for_damping(damping)
        accel = bool(rng.integers(0, 2))
        tol = float(rng.choice([0.0, 1.5e-6]))
        k_f = int(rng.choice([1, 3, 5, 5, 10, 16]))
        seed = int(rng.integers(1, 1 << 30))
        if forced is not None:
            v, e, dim, power_law, pfrac, b, damping, iters, accel, tol, k_f, seed = (forced[k] for k in (
                "v", "e", "dim", "power_law", "pfrac", "b", "damping", "iters", "accel", "tol", "k_f", "seed"))
        par = dict(v=v, e=e, dim=dim, power_law=power_law, pfrac=pfrac, b=b, damping=damping, iters=iters, accel=accel,
                   tol=tol, k_f=k_f, seed=seed)
        rng_case = np.random.default_rng(seed ^ 0x5EED)      # filter outcomes, k, engine knobs: a function of the case alone
        # engine knobs that must not change any result beyond the parity bars: vertex numbering, long-row cut, tuning bits
        locality = [None, None, "auto", "degree"][int(rng_case.integers(0, 4))]
        seg_len = int(rng_case.choice([0, 0, 64, 128, 512]))
        tuning = 0
        for bit in (_lib.OPT_NT_CSR, _lib.OPT_NT_STORE, _lib.OPT_TEMPORAL16, _lib.OPT_SLABS_PER_WG_1, _lib.OPT_XCD_BLOCKED):
            if rng_case.random() < 0.15:
                tuning |= bit
        pw = float(rng_case.choice([0.05, 0.05, 0.05, 0.01, 0.5, 1.0]))       # passage_node_weight (reference default 0.05)
        emb_kind = ["bf16", "bf16", "f32", "fp16"][int(rng_case.integers(0, 4))]     # f32: HRAG_F32_SPLIT (the mirror's default)
        par.update(locality=locality, seg_len=seg_len, tuning=tuning, pw=pw, emb=emb_kind)
        kg, pass_bits, fact_bits, index = make_case(v, e, dim, seed=seed, passage_frac=pfrac, power_law=power_law)
        index = dataclasses.replace(index, damping=damping, linking_top_k=k_f, passage_node_weight=pw)
        n_p = kg.n_passages
        k_docs = int(min(n_p, rng_case.choice([200, 2048] if args.big else [10, 100, 500])))
        qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=seed + 5)
        qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=seed + 6)
        qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
        pass_arg, fact_arg, to_dev = pass_bits, fact_bits, (lambda bits, x: bf16(bits, dev))
        if emb_kind != "bf16":
            # rows / queries that are NOT bf16-representable: fp32 (split into hi + lo fp16 planes by the engine) or fp16
            def jitter(x, sd):
                y = x + np.random.default_rng(sd).standard_normal(x.shape).astype(np.float32) * np.float32(2e-3)
                return (y / np.linalg.norm(y, axis=1, keepdims=True)).astype(np.float32)
            pe, fe = jitter(index.passage_emb, seed + 11), jitter(index.fact_emb, seed + 12)
            qf, qp = jitter(qf, seed + 13), jitter(qp, seed + 14)
            if emb_kind == "fp16":
                pe, fe = pe.astype(np.float16), fe.astype(np.float16)
                qf, qp = qf.astype(np.float16).astype(np.float32), qp.astype(np.float16).astype(np.float32)
            pass_arg, fact_arg = pe, fe
            index = dataclasses.replace(index, passage_emb=pe.astype(np.float32), fact_emb=fe.astype(np.float32))
            tdt = torch.float16 if emb_kind == "fp16" else torch.float32
            to_dev = lambda bits, x: torch.from_numpy(x).to(dev).to(tdt)
        # what the recognition-memory filter keeps: all / a random subset / nothing (-> DPR ranking)
        keep_mode = rng_case.integers(0, 3, b)
        flags_engine = (_lib.OPT_ACCEL if accel else 0) | tuning
        try:
            with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_arg, fact_arg, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                                max_batch=b, max_topk=k_docs, flags=flags_engine, locality=locality, sell_seg_len=seg_len) as eng:
                idx, sc = eng.score_facts(to_dev(qf_bits, qf), k=k_f)
                idx_h, sc_h = idx.cpu().numpy(), sc.cpu().numpy()
                kept_idx = np.full((b, k_f), -1, np.int32)
                kept_sc = np.zeros((b, k_f), np.float32)
                cnt = np.zeros(b, np.int32)
                for q in range(b):
                    valid = [j for j in range(k_f) if idx_h[q, j] >= 0]
                    if keep_mode[q] == 1:
                        valid = [j for j in valid if rng_case.random() < 0.6]
                    elif keep_mode[q] == 2:
                        valid = []
                    kept_idx[q, :len(valid)] = idx_h[q, valid]
                    kept_sc[q, :len(valid)] = sc_h[q, valid]
                    cnt[q] = len(valid)
                fn = eng.retrieve_converged if tol > 0 else eng.retrieve
                kw = dict(link_top_k=k_f, damping=damping, passage_node_weight=pw, ppr_iters=iters, k=k_docs)
                if tol > 0:
                    kw.update(ppr_tol=tol, ppr_max_iters=400)
                out = fn(to_dev(qp_bits, qp), torch.from_numpy(kept_idx), torch.from_numpy(kept_sc), torch.from_numpy(cnt), **kw)
                torch.cuda.synchronize()
                d_idx, d_sc, fl = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy(), out.flags.cpu().numpy()
                used = int(out.iters_used.max()) if out.iters_used is not None else iters
                width = eng.timings()["slab_width"]
                # the run_ppr seam: arbitrary reset vectors (a few of them), every vertex's score, caller's numbering
                nb = min(b, int(rng_case.choice([1, 3, 8, 12])))
                reset = np.zeros((nb, kg.num_vertices), np.float32)
                for i in range(nb):
                    hot = rng_case.choice(kg.num_vertices, int(rng_case.integers(1, 40)), replace=False)
                    reset[i, hot] = rng_case.random(len(hot)).astype(np.float32) + 1e-3
                seam_iters = int(rng_case.choice([iters, 2 * iters]))
                xs, xfl = eng.ppr(torch.from_numpy(reset), damping, seam_iters)
                xs, xfl = xs.cpu().numpy(), xfl.cpu().numpy()
        except Exception as exc:  # noqa: BLE001
            bad += 1
            par["error"] = f"{type(exc).__name__}: {exc}"
            print("FAIL", json.dumps(par), flush=True)
            cases.append(par)
            continue
        worst, ok = 0.0, True
        why = ""
        ok_a, why_a = True, ""
        for q in sorted(set(np.linspace(0, b - 1, min(b, 3)).astype(int).tolist())):      # phase A against the oracle
            fs = oracle.fact_scores(index.fact_emb, qf[q])
            cand = oracle.topk_desc(fs, k_f)
            nf = min(k_f, len(fs))
            same = tie_aware_equal(idx_h[q][:nf], cand[:nf], fs[cand[:nf]], rel_gap=0.0, abs_gap=4e-6)
            devf = float(np.abs(sc_h[q][:nf].astype(np.float64) - fs[idx_h[q][:nf]]).max())
            if not same or devf > 2e-6 or np.any(idx_h[q][nf:] != -1):
                ok_a, why_a = False, f"q{q} fact top-k: ids {same}, abs score dev {devf:.2e}"
        for i in range(nb):
            want = oracle.ppr_power(index.p, reset[i].astype(np.float64), damping, seam_iters)
            nzw = want > 1e-12                      # the bar of tests/test_gpu_locality.py: rtol 2e-5, atol 1e-12
            seam = float(np.abs(xs[i][nzw] / want[nzw] - 1).max()) if nzw.any() else 0.0
            if seam >= 2e-5 or xfl[i] != 0 or not np.all(xs[i][want == 0] == 0):
                ok, why = False, f"ppr seam vector {i}: rel dev {seam:.2e} flags {int(xfl[i])}"
        for q in sorted(set(np.linspace(0, b - 1, min(b, 2 if args.big else 4)).astype(int).tolist())):
            if cnt[q] == 0:
                if not (fl[q] & 1):
                    ok, why = False, f"q{q}: no DPR flag"
                # min-max normalised scores in [0, 1]: an ABSOLUTE bar (2e-6, as tests/test_ref_golden.py) -- the relative
                # error of a score next to the minimum is the reference's own fp32 dot noise
                ids, scs = oracle.retrieve_dpr_one(index, qp[q])
                full = np.empty(n_p)
                full[ids] = scs
                dev_abs = float(np.abs(d_sc[q].astype(np.float64) - full[d_idx[q]]).max())
                same = tie_aware_equal(d_idx[q], ids[:len(d_idx[q])], scs[:len(d_idx[q])], rel_gap=0.0, abs_gap=4e-6)
                if dev_abs > 2e-6 or not same:
                    ok, why = False, f"q{q} (DPR ranking): abs score dev {dev_abs:.2e}, ids {same}"
                continue
            else:
                scores = np.zeros(len(index.subj_vertex), np.float32)
                kept = kept_idx[q, :cnt[q]]
                scores[kept] = kept_sc[q, :cnt[q]]
                try:
                    sid, sw = oracle.seed_weights(index, scores, kept.tolist(), k_f)
                except AssertionError:          # the reference's asserts (:1541, :1644): the engine must flag, not rank
                    if not (fl[q] & 6):
                        ok, why = False, f"q{q}: the oracle asserts, flags {int(fl[q])}"
                    continue
                dpr_ids, dpr_sc = oracle.dense_passage_scores(index.passage_emb, qp[q])
                by_p = np.empty_like(dpr_sc)
                by_p[dpr_ids] = dpr_sc
                ids, scs, x = oracle.run_ppr(index, oracle.reset_vector(index, sid, sw, by_p), damping, "exact", iters)
                full = x[index.passage_vertex]
                rep = ranked_parity(d_idx[q], d_sc[q], ids, scs, full)
                # fixed sweep counts are only as accurate as the graph mixes (tol = 0): compare with the same count
                if tol == 0.0 and rep["worst_rel_err"] >= 1e-5:
                    ids, scs, x = oracle.run_ppr(index, oracle.reset_vector(index, sid, sw, by_p), damping, "power", iters)
                    full = x[index.passage_vertex]
                    rep = ranked_parity(d_idx[q], d_sc[q], ids, scs, full)
            allow = float(prior_noise_allowance(index, qp[q])[d_idx[q]].max()) if cnt[q] else 0.0
            worst = max(worst, rep["worst_rel_err"])
            if rep["worst_rel_err"] >= 1e-5 + allow or (not rep["equal"] and rep["worst_rel_err"] < 1e-5 and allow < 1e-6):
                ok, why = False, f"q{q}: err {rep['worst_rel_err']:.2e} (allow {allow:.1e}) equal={rep['equal']} gap={rep['rel_gap']:.1e}"
            if fl[q] & ~1:
                ok, why = False, f"q{q}: flags {int(fl[q])}"
        if not ok_a:
            ok, why = False, why_a
        par.update(worst=worst, ok=ok, sweeps=used, width=width, k=k_docs)
        if args.verbose:
            par.update(flags=sorted(set(int(f) for f in fl)), flagged_queries=[int(q) for q in np.flatnonzero(fl & ~1)][:16],
                       kept_counts=cnt.tolist()[:16])
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
