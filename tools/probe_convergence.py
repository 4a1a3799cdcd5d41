"""GPU probe (round 3): the convergence contract of hrag_retrieve on the adversarial graphs of
tests/test_gpu_fp8_adversarial.py -- error vs the exact solution, reported residual, sweeps used."""
import dataclasses
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from hipporag_amd import synth  # noqa: E402
from hipporag_amd.engine import HippoRAGEngine  # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float  # noqa: E402
import test_gpu_fp8_adversarial as adv  # noqa: E402

dev = torch.device("cuda", 0)
out = {}
for name, (make, damping, iters) in sorted(adv.CASES.items()):
    n, src, dst, w, pv, pinned = make()
    csr, pass_bits, fact_bits, index = adv._index(n, src, dst, w, pv, 64, seed=11, pinned_facts=pinned)
    index = dataclasses.replace(index, damping=damping)
    n_p = len(pv)
    b = 65
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    for i in range(len(pinned)):
        qf_bits[i] = fact_bits[i]
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    check = list(range(0, b, 6))
    exact = {q: oracle.retrieve_one(index, qf[q], qp[q]).x[index.passage_vertex] for q in check}
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(adv._bf16(qf_bits, dev), k=5)
        cnt = adv._t(np.full(b, 5, np.int32), dev)
        for tol, mx in ((0.0, 0), (3e-6, 29), (1e-7, 29)):
            o = eng.retrieve(adv._bf16(qp_bits, dev), idx, sc, cnt, damping=damping, ppr_iters=iters, k=n_p,
                             ppr_tol=tol, ppr_max_iters=mx)
            torch.cuda.synchronize()
            gi, gs = o.doc_idx.cpu().numpy(), o.doc_score.cpu().numpy()
            fl, rs, used = o.flags.cpu().numpy(), o.residual.cpu().numpy(), o.iters_used.cpu().numpy()
            errs = []
            for q in check:
                full = np.empty(n_p)
                full[gi[q]] = gs[q]
                want = exact[q]
                nz = want > 0
                errs.append(float(np.abs(full[nz] / want[nz] - 1).max()))
            errs = np.array(errs)
            rec = dict(err_max=float(errs.max()), resid_max=float(rs.max()), resid_checked=[float(rs[q]) for q in check],
                       err_checked=errs.tolist(), iters_used=int(used.max()), flags=sorted(set(int(f) for f in fl)),
                       slab_width=eng.timings()["slab_width"])
            out[f"{name} tol={tol}"] = rec
            print(name, tol, json.dumps({k: v for k, v in rec.items() if "checked" not in k}),
                  "min est/err", float(np.min(np.array(rec["resid_checked"]) / np.maximum(errs, 1e-12))), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_conv_probe.json"), "w"), indent=1)
