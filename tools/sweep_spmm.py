#!/usr/bin/env python
"""A/B sweep of the PPR SpMM kernel variants on one GPU (one process, interleaved rounds).

    python tools/sweep_spmm.py --config cfg3 --out gpurun_out/sweep.json

For every variant (slab width x long-row threshold x flags) an engine is built on the same graph,
the state is filled by a real PPR run, and `hrag_ppr_sweeps(main_only)` is timed with HIP events.
"""

import argparse
import json
import os
import sys
import time

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
    ap.add_argument("--out", default="gpurun_out/sweep.json")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--variants", default="")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    V, E, B, seed = cfg["V"], cfg["E"], cfg["B"], cfg["seed"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.time()
    kg = synth.make_kg(V, E, seed)
    print(f"graph built in {time.time() - t0:.1f}s nnz={kg.csr.nnz}", flush=True)
    pemb = synth.make_embeddings_torch(kg.n_passages, 64, 1, dev)
    femb = synth.make_embeddings_torch(kg.n_facts, 64, 2, dev)

    # HBM copy ceiling on this box (float4 copy of 1 GiB)
    a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    print(f"torch copy: {copy_gbs:.0f} GB/s (read+write)", flush=True)

    variants = [
        dict(name="bc32", slab=32, long=0, seg=0, flags=0),
        dict(name="bc16", slab=16, long=0, seg=0, flags=0),
        dict(name="bc64", slab=64, long=0, seg=0, flags=0),
        dict(name="bc8", slab=8, long=0, seg=0, flags=0),
        dict(name="bc32_natural", slab=32, long=0, seg=0, flags=1),
        dict(name="bc32_ntcsr", slab=32, long=0, seg=0, flags=2),
        dict(name="bc32_ntst", slab=32, long=0, seg=0, flags=4),
        dict(name="bc32_nt_both", slab=32, long=0, seg=0, flags=6),
        dict(name="bc32_short32", slab=32, long=32, seg=0, flags=0),
        dict(name="bc32_short128", slab=32, long=128, seg=0, flags=0),
        dict(name="bc32_seg256", slab=32, long=0, seg=256, flags=0),
        dict(name="bc64_nt_both", slab=64, long=0, seg=0, flags=6),
    ]
    if args.variants:
        keep = set(args.variants.split(","))
        variants = [v for v in variants if v["name"] in keep]
    qf, _ = synth.make_queries_torch(femb, B, 7)
    qp, _ = synth.make_queries_torch(pemb, B, 8)
    cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
    alg = spmm_algorithmic_bytes(kg.csr.nnz, V, kg.n_passages, B)
    results = {v["name"]: dict(v, main_ms=[], all_ms=[]) for v in variants}
    ref_idx = None
    for rnd in range(args.rounds):
        for v in variants:
            eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                                 max_batch=B, max_topk=200, slab_width=v["slab"], long_row_nnz=v["long"],
                                 segment_nnz=v["seg"], flags=v["flags"])
            idx, sc = eng.score_facts(qf, k=5)
            out = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)      # real state in x / tele / seeds
            torch.cuda.synchronize()
            if ref_idx is None:
                ref_idx = out.doc_idx.clone()
            same = bool(torch.equal(ref_idx, out.doc_idx))
            for key, main_only in (("main_ms", True), ("all_ms", False)):
                eng.ppr_sweeps(B, 2, 0.5, main_only=main_only)
                e0.record()
                eng.ppr_sweeps(B, args.launches, 0.5, main_only=main_only)
                e1.record()
                torch.cuda.synchronize()
                results[v["name"]][key].append(e0.elapsed_time(e1) / args.launches)
            results[v["name"]]["ids_equal_to_first_variant"] = same
            results[v["name"]]["n_long"] = eng.timings()["n_long_rows"]
            eng.close()
            del eng
            torch.cuda.empty_cache()
            r = results[v["name"]]
            print(f"round {rnd} {v['name']:>16}: main {r['main_ms'][-1]:8.3f} ms  all {r['all_ms'][-1]:8.3f} ms  "
                  f"{alg / (r['main_ms'][-1] * 1e-3) / 1e9:7.0f} GB/s alg  ids_same={same} n_long={r['n_long']}",
                  flush=True)
    summary = {"config": args.config, "B": B, "algorithmic_bytes": alg, "copy_gbs": copy_gbs, "variants": results}
    for name, r in results.items():
        r["main_ms_median"] = float(np.median(r["main_ms"]))
        r["all_ms_median"] = float(np.median(r["all_ms"]))
        r["alg_gbs"] = alg / (r["main_ms_median"] * 1e-3) / 1e9
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(summary, open(args.out, "w"), indent=1)
    print(json.dumps({k: (round(r["main_ms_median"], 3), round(r["alg_gbs"])) for k, r in results.items()}))


if __name__ == "__main__":
    main()
