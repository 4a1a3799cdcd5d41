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
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    keys = rng.standard_normal((args.n, args.dim)).astype(np.float32)
    q = keys[: args.nq]
    retrieve_knn(None, None, q[:256], keys[:5000], k=16, return_arrays=True)        # warm-up / module load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, sc = retrieve_knn(None, None, q, keys, k=args.k, return_arrays=True)
    torch.cuda.synchronize()
    t_ours = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx1, sc1 = retrieve_knn(None, None, q, keys, k=args.k, precision="bf16", return_arrays=True)
    torch.cuda.synchronize()
    t_ours1 = time.perf_counter() - t0
    reference_style(q[:1000], keys[:20000], 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ridx, rsc = reference_style(q, keys, args.k)
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    flops = 2.0 * args.nq * args.n * args.dim
    same = float((idx == ridx).mean())
    print(json.dumps({"n_keys": args.n, "n_queries": args.nq, "dim": args.dim, "k": args.k,
                      "ours_bf16x3_s": t_ours, "ours_bf16_s": t_ours1, "torch_fp32_reference_style_s": t_ref,
                      "speedup_bf16x3": t_ref / t_ours, "effective_tflops_bf16x3": 3 * flops / t_ours / 1e12,
                      "ids_equal_fraction": same, "max_abs_score_diff": float(np.abs(sc - rsc).max())}))


if __name__ == "__main__":
    main()
