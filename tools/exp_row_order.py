#!/usr/bin/env python
"""Experiment (SURVEY.md section 7 / DESIGN.md section 4): does the ORDER in which the SELL-8 rows are processed
change the L2 reuse of the PPR gathers?  Rows of equal length can be permuted freely; rows that share
in-neighbours, processed close together on one XCD, could hit each other's lines in that XCD's 4 MB L2.

    python tools/exp_row_order.py            # times the mode-C sweep under the three row orders
    HRAG_FLAGS=64|128 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ... python tools/pmc_target.py   (tools/gpu_r02_d.sh)

Row orders (hrag_opts.flags): 0 = length only (ties in vertex order), HRAG_OPT_ROWS_BY_MINCOL = 64,
HRAG_OPT_ROWS_BFS = 128.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bench import CONFIGS
from hipporag_amd import synth
from hipporag_amd.engine import HippoRAGEngine


def main():
    cfg = CONFIGS[os.environ.get("HRAG_PMC_CONFIG", "cfg3")]
    V, E, B, seed = cfg["V"], cfg["E"], cfg["B"], cfg["seed"]
    dev = torch.device("cuda", 0)
    kg = synth.make_kg(V, E, seed, power_law=bool(cfg.get("power_law")))
    pemb = synth.make_embeddings_torch(kg.n_passages, 64, 1, dev)
    femb = synth.make_embeddings_torch(kg.n_facts, 64, 2, dev)
    qf, _ = synth.make_queries_torch(femb, B, 7)
    qp, _ = synth.make_queries_torch(pemb, B, 8)
    cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
    out = {"workload": cfg["label"], "kernel": "ppr8_kernel mode C", "launch_ms": {}}
    ref = None
    for name, flags in (("length_only", 0), ("by_min_column", 64), ("bfs_rank", 128)):
        with HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                            max_batch=B, max_topk=200, flags=flags) as eng:
            idx, sc = eng.score_facts(qf, k=5)
            res = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
            torch.cuda.synchronize()
            if ref is None:
                ref = res.doc_idx.clone()
            out.setdefault("same_top200_ids_as_length_only", {})[name] = bool(torch.equal(ref, res.doc_idx))
            eng.ppr_sweeps(B, 4, 0.5, main_only=True, f8=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.ppr_sweeps(B, 40, 0.5, main_only=True, f8=True)
            e1.record()
            torch.cuda.synchronize()
            out["launch_ms"][name] = e0.elapsed_time(e1) / 40
    print(json.dumps(out))


if __name__ == "__main__":
    main()
