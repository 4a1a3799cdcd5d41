#!/usr/bin/env python
"""A/B timing of the fp16-state PPR sweep (ppr16_kernel) under engine flag variants, one process.

    python tools/sweep_ppr16.py --config cfg3 [--batches 64,128,256]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from bench import CONFIGS, spmm_algorithmic_bytes
from hipporag_amd import synth
from hipporag_amd.engine import HippoRAGEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--batches", default="")
    ap.add_argument("--flags", default="0,16")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--out", default="gpurun_out/sweep_ppr16.json")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    V, E, seed = cfg["V"], cfg["E"], cfg["seed"]
    batches = [int(b) for b in args.batches.split(",")] if args.batches else [cfg["B"]]
    dev = torch.device("cuda", 0)
    kg = synth.make_kg(V, E, seed)
    pemb = synth.make_embeddings_torch(kg.n_passages, 64, 1, dev)
    femb = synth.make_embeddings_torch(kg.n_facts, 64, 2, dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for rnd in range(args.rounds):
        for B in batches:
            for fl in [int(f) for f in args.flags.split(",")]:
                eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex,
                                     kg.num_chunks, max_batch=B, max_topk=200, flags=fl | 32)   # 32 = HRAG_OPT_NO_FP8: this tool measures the fp16 path
                qf, _ = synth.make_queries_torch(femb, B, 7)
                qp, _ = synth.make_queries_torch(pemb, B, 8)
                cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
                idx, sc = eng.score_facts(qf, k=5)
                eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
                torch.cuda.synchronize()
                key = f"B{B}_flags{fl}"
                for name, main_only in (("main_ms", True), ("all_ms", False)):
                    eng.ppr_sweeps(B, 2, 0.5, main_only=main_only, f16=True)
                    e0.record()
                    eng.ppr_sweeps(B, args.launches, 0.5, main_only=main_only, f16=True)
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault(key, {}).setdefault(name, []).append(e0.elapsed_time(e1) / args.launches)
                eng.close()
                del eng
                torch.cuda.empty_cache()
                alg = spmm_algorithmic_bytes(kg.csr.nnz, V, kg.n_passages, B, 2)
                r = res[key]
                print(f"round {rnd} {key:>16}: main {r['main_ms'][-1]:7.3f} ms  all {r['all_ms'][-1]:7.3f} ms  "
                      f"alg {alg / (r['main_ms'][-1] * 1e-3) / 1e9:6.0f} GB/s  gather "
                      f"{kg.csr.nnz * ((B + 63) // 64) * 128 / (r['main_ms'][-1] * 1e-3) / 1e9:6.0f} GB/s", flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump({k: {n: float(np.median(v)) for n, v in r.items()} for k, r in res.items()}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
