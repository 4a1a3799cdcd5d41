#!/usr/bin/env python
"""Index-time KNN: hipporag_amd.knn.retrieve_knn vs the reference algorithm (blocked torch.mm + torch.topk,
embed_utils.py:6-94) on the same GPU.   python tools/bench_knn.py --n 100000 --dim 768 --k 2047"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from hipporag_amd.knn import retrieve_knn


def reference_style(q, keys, k, qb=1000, kb=10000):
    dev = torch.device("cuda")
    qv = torch.nn.functional.normalize(torch.from_numpy(q), dim=1)
    kv = torch.nn.functional.normalize(torch.from_numpy(keys), dim=1)
    out_i, out_s = [], []
    for i in range(0, len(qv), qb):
        qq = qv[i:i + qb].to(dev)
        ss, ii, off = [], [], 0
        for j in range(0, len(kv), kb):
            kk = kv[j:j + kb].to(dev)
            sim = torch.mm(qq, kk.T)
            s, idx = torch.topk(sim, min(k, kk.size(0)), dim=1, largest=True, sorted=True)
            ss.append(s); ii.append(idx + off); off += kk.size(0)
        ss, ii = torch.cat(ss, 1), torch.cat(ii, 1)
        s, pos = torch.topk(ss, min(k, ss.size(1)), dim=1, largest=True, sorted=True)
        out_s.append(s.cpu()); out_i.append(torch.gather(ii, 1, pos).cpu())
    return torch.cat(out_i).numpy(), torch.cat(out_s).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--nq", type=int, default=20000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=2047)
    ap.add_argument("--ref-queries", type=int, default=20000, help="queries of the torch fp32 reference-style leg (0: skip)")
    ap.add_argument("--query-batch", type=int, default=1000)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    keys = rng.standard_normal((args.n, args.dim)).astype(np.float32)
    q = keys[: args.nq]
    retrieve_knn(None, None, q[:256], keys[:5000], k=16, return_arrays=True)        # warm-up / module load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, sc = retrieve_knn(None, None, q, keys, k=args.k, query_batch_size=args.query_batch, return_arrays=True)
    torch.cuda.synchronize()
    t_ours = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx1, sc1 = retrieve_knn(None, None, q, keys, k=args.k, precision="bf16", query_batch_size=args.query_batch,
                             return_arrays=True)
    torch.cuda.synchronize()
    t_ours1 = time.perf_counter() - t0
    # the call add_synonymy_edges needs (HippoRAG.py:985-1018): neighbours above the 0.8 threshold, at most 100 + self
    t0 = time.perf_counter()
    idx_s, sc_s = retrieve_knn(None, None, q, keys, k=103, query_batch_size=args.query_batch, return_arrays=True,
                               min_score=0.8)
    torch.cuda.synchronize()
    t_syn = time.perf_counter() - t0
    flops = 2.0 * args.nq * args.n * args.dim
    out = {"n_keys": args.n, "n_queries": args.nq, "dim": args.dim, "k": args.k,
           "ours_f32_s": t_ours, "ours_bf16_s": t_ours1, "ours_f32_synonymy_call_s": t_syn,
           "synonymy_call": "k = 103, min_score = 0.8: what add_synonymy_edges reads; useful rate "
                            f"{flops / t_syn / 1e12:.1f} TFLOP/s (2 * nq * n * dim per second).  Since round 6 the first pass "
                            "runs over the hi . qhi third of the split layout (hrag_sim_topk_min_score), so the MFMA work "
                            "EXECUTED is ~1x that, not 3x; at the 3x of the exact pass the call would read "
                            f"{3 * flops / t_syn / 1e12:.1f} TFLOP/s-equivalent",
           "extrapolated_full_self_knn_synonymy_s": t_syn * args.n / args.nq,
           "useful_tflops_f32": flops / t_ours / 1e12,
           "roofline": {"bound": "mfma", "achieved": flops / t_syn / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": flops / t_syn / 1e12 / 2500.0,
                        "note": "the synonymy call (what index() needs): MFMA work executed (first pass over dim elements "
                                "per product; the rescoring of the few tiles above the threshold is negligible) over the "
                                "WHOLE call -- host -> device copy and split of the keys, per-block GEMM with tile "
                                "maxima, tile selection and rescoring, result copy -- against the dense fp16 peak; rounds "
                                "3 - 5 quoted 3 * dim per product (the exact first pass they ran)"},
           "roofline_full_lists": {"bound": "mfma", "achieved": 3 * flops / t_ours / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                   "frac": 3 * flops / t_ours / 1e12 / 2500.0,
                                   "note": "k = 2047 full lists: dominated by the exact top-2047 of every [1, n_keys] score "
                                           "row (row_topk_kernel) and the 16 KB per query that cross PCIe"},
           "extrapolated_full_self_knn_s": t_ours * args.n / args.nq}
    nr = min(args.ref_queries, args.nq)
    if nr > 0:
        reference_style(q[:1000], keys[:20000], 16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ridx, rsc = reference_style(q[:nr], keys, args.k)
        torch.cuda.synchronize()
        t_ref = (time.perf_counter() - t0) * args.nq / nr
        out.update({"torch_fp32_reference_style_s": t_ref, "reference_queries_timed": nr, "speedup_f32": t_ref / t_ours,
                    "ids_equal_fraction": float((idx[:nr] == ridx).mean()),
                    "max_abs_score_diff": float(np.abs(sc[:nr] - rsc).max())})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
