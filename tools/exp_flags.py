#!/usr/bin/env python
"""A/B tool: the fp8 PPR sweeps of cfg 3 under different hrag_opts.flags values (one engine each).

    python tools/exp_flags.py 0 256          # e.g. HRAG_OPT_SLABS_PER_WG_1
Prints one JSON line: launch time of the mode-C / mode-B (3-byte residual) sweeps and of one whole retrieve per
flag value, and whether the ranked ids / scores are bit-identical to the first flag value's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bench import CONFIGS
from hipporag_amd import synth
from hipporag_amd.engine import HippoRAGEngine


def timed(fn, n):
    fn(4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    flag_values = [int(a, 0) for a in sys.argv[1:]] or [0]
    cfg = CONFIGS[os.environ.get("HRAG_PMC_CONFIG", "cfg3")]
    V, E, seed = cfg["V"], cfg["E"], cfg["seed"]
    B = int(os.environ.get("HRAG_PMC_BATCH", cfg["B"]))
    dev = torch.device("cuda", 0)
    kg = synth.make_kg(V, E, seed, power_law=bool(cfg.get("power_law")))
    pemb = synth.make_embeddings_torch(kg.n_passages, 64, 1, dev)
    femb = synth.make_embeddings_torch(kg.n_facts, 64, 2, dev)
    qf, _ = synth.make_queries_torch(femb, B, 7)
    qp, _ = synth.make_queries_torch(pemb, B, 8)
    cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
    out = {"workload": cfg["label"], "batch": B, "flags": {}}
    ref = None
    for flags in flag_values:
        with HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                            max_batch=B, max_topk=200, flags=flags) as eng:
            idx, sc = eng.score_facts(qf, k=5)
            res = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
            torch.cuda.synchronize()
            if ref is None:
                ref = (res.doc_idx.clone(), res.doc_score.clone())
            r = {"same_as_first": bool(torch.equal(ref[0], res.doc_idx) and torch.equal(ref[1], res.doc_score))}
            r["mode_C_ms"] = timed(lambda n: eng.ppr_sweeps(B, n, 0.5, f8=True, f8_mode="C"), 40)
            r["mode_B3_ms"] = timed(lambda n: eng.ppr_sweeps(B, n, 0.5, f8=True, f8_mode="B", f8_rio=3), 20)
            r["retrieve_ms"] = timed(lambda n: [eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200) for _ in range(n)], 5)
            out["flags"][str(flags)] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
