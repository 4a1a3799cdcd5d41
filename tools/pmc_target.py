#!/usr/bin/env python
"""Small target for the rocprofv3 PMC passes: build the cfg-3 engine, fill the PPR state with one
real retrieve, then run a few sweeps of the dominant kernel.  Run it once per counter:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python tools/pmc_target.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python tools/pmc_target.py
then   python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > profiles/pmc_traffic.json
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bench import CONFIGS
from hipporag_amd import synth
from hipporag_amd.engine import HippoRAGEngine

cfg = CONFIGS[os.environ.get("HRAG_PMC_CONFIG", "cfg3")]
V, E, B, seed = cfg["V"], cfg["E"], int(os.environ.get("HRAG_PMC_BATCH", cfg["B"])), cfg["seed"]
dev = torch.device("cuda", 0)
if cfg.get("real2wiki"):      # the real-topology graph (tools/real2wiki.py), HRAG_PMC_CONFIG=real2wiki
    from tools import real2wiki as rw
    kg = rw.build_kg(int(cfg["tiles"]))
else:
    kg = synth.make_kg(V, E, seed, community=int(os.environ.get("HRAG_COMMUNITY", cfg.get("community", 0))))
pemb = synth.make_embeddings_torch(kg.n_passages, 64, 1, dev)
femb = synth.make_embeddings_torch(kg.n_facts, 64, 2, dev)
eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                     max_batch=B, max_topk=200, slab_width=int(os.environ.get("HRAG_SLAB", "0")),
                     flags=int(os.environ.get("HRAG_FLAGS", "0")), sell_sigma=int(os.environ.get("HRAG_SELL_SIGMA", "0")),
                     locality=os.environ.get("HRAG_LOCALITY") or None)
qf, _ = synth.make_queries_torch(femb, B, 7)
qp, _ = synth.make_queries_torch(pemb, B, 8)
cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
idx, sc = eng.score_facts(qf, k=5)
eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)   # fp8 state for B > 64, fp16 for B > 8, small-batch below
width = eng.timings()["slab_width"]
if width == 128:      # staged fp8 state: every instantiation of ppr8_kernel (template argument = Ppr8Mode: C 0, B 1, F 2, B0 3)
    # ... and the residual forms of B / F (second template argument: 0 fp32, 2 fp32 in / 3-byte out, 3 3-byte, F: 1)
    for mode, rio in (("C", 0), ("B0", 0), ("B", 0), ("B", 2), ("B", 3), ("F", 1)):
        eng.ppr_sweeps(B, 4, 0.5, main_only=True, f8=True, f8_mode=mode, f8_rio=rio)
    if B > 128:   # the gathers of a stage sweep alone (round 6: ppr8_pair_replay_kernel, bench.py roofline.gather_replay_ms)
        eng.ppr_sweeps(B, 4, 0.5, f8=True, f8_gather_replay=True)
else:
    eng.ppr_sweeps(B, 4, 0.5, main_only=True, f16=width == 64 and B > 8, small=B <= 8 and width <= 8)
torch.cuda.synchronize()
eng.close()
print("pmc target done")
