#!/usr/bin/env python
"""Randomised soak of the multi-GPU ORCHESTRATION over a real process group (CPU, gloo): world sizes 2 .. 4, random index
sizes, batches (ragged slabs), exchange-group counts and seeds; every rank is a process running dist.ShardedRetriever /
dist.HybridRetriever over dist.TorchComm (the class the RCCL job uses; gloo instead of nccl) with the numpy stand-in engines
of tests/test_shard_orchestration.py, checked against the single-process oracle, all ranks agreeing.  What is under test:
the collectives' call pattern (in-place all-gather of the owners' blocks per exchange group and sweep, min / max / sum
all-reduces, candidate merge, the hybrid's all-to-all of score rows) for world sizes and shapes the fixed tests do not visit.

    python tools/soak_gloo.py [--cases 30] [--seed 1]"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, par, ret):
    import torch
    import torch.distributed as dist
    import oracle
    from hipporag_amd import dist as hd, synth
    from hipporag_amd.graph import bf16_bits_to_float
    from tests import test_shard_orchestration as tso
    from tests.helpers import make_case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kg, pass_bits, fact_bits, index = make_case(par["v"], par["e"], 32, seed=par["seed"], power_law=par["power_law"])
        sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
        b = par["b"]
        qf = torch.from_numpy(bf16_bits_to_float(synth.make_queries_np(fact_bits, b, par["seed"] + 1)[0]))
        qp = torch.from_numpy(bf16_bits_to_float(synth.make_queries_np(pass_bits, b, par["seed"] + 2)[0]))
        k = min(40, kg.n_passages)
        rng = np.random.default_rng(par["seed"] + 3)
        cnt = torch.from_numpy(rng.choice([0, 2, 5, 5, 5], b).astype(np.int32))
        if par["mode"] == "rowshard":
            rs = hd.ShardedRetriever(tso.FakeShardStages(sidx, index, rank), hd.TorchComm(rank, world), groups=par["groups"])
            idx, sc = rs.score_facts(qf, k=5)
            d_idx, d_sc, flags = rs.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=k)[:3]
            rows = range(b)
        else:
            sim = tso.FakeShardStages(sidx, index, rank)
            from types import SimpleNamespace
            sim.e = SimpleNamespace(sim_scores=lambda which, q: torch.from_numpy((q.double().numpy() @ sim.pe.T).astype(np.float32)))
            hy = hd.HybridRetriever(sim, tso._FakePprEngine(index), hd.TorchComm(rank, world), sidx.passages)
            idx, sc = hy.score_facts(qf, k=5)
            out = hy.retrieve(qp, idx, sc, cnt, link_top_k=5, damping=0.5, passage_node_weight=0.05, ppr_iters=20, k=k)
            d_idx, d_sc, flags = out.doc_idx, out.doc_score, getattr(out, "flags", None)   # the stand-in PPR engine reports none
            mine = hy.my_rows(b)
            rows = range(mine.start, mine.stop)
        check = list(rows)[:: max(1, len(rows) // 5)]
        for i, q in enumerate(rows):
            if q not in check:
                continue
            j = i if par["mode"] == "hybrid" else q
            flt = (lambda cand, n=int(cnt[q]): cand[:n])
            ref = oracle.retrieve_one(index, qf[q].numpy(), qp[q].numpy(), filter_fn=flt, ppr_mode="power", ppr_iters=20)
            np.testing.assert_array_equal(idx[q].numpy(), ref.fact_candidates)
            assert flags is None or bool(int(flags[j]) & 1) == ref.used_dpr
            np.testing.assert_array_equal(d_idx[j].numpy(), ref.sorted_doc_ids[:k])
            np.testing.assert_allclose(d_sc[j].numpy(), ref.sorted_doc_scores[:k], rtol=3e-6, atol=1e-7)
        if par["mode"] == "rowshard":                                # the replicated answer is the same on every rank
            gathered = [torch.empty_like(d_idx) for _ in range(world)]
            dist.all_gather(gathered, d_idx)
            assert all(torch.equal(g, d_idx) for g in gathered)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch.multiprocessing as mp
    rng = np.random.default_rng(args.seed)
    bad = n = 0
    for n in range(1, args.cases + 1):
        world = int(rng.choice([2, 3, 4]))
        mode = str(rng.choice(["rowshard", "rowshard", "hybrid"]))
        b = int(rng.choice([7, 33, 70, 130]))
        if mode == "hybrid":
            b = max(world, b // world * world)
        par = dict(world=world, mode=mode, b=b, groups=int(rng.choice([0, 1, 2, 3])), v=int(rng.choice([600, 1500, 4000])),
                   e=0, power_law=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
        par["e"] = par["v"] * int(rng.choice([4, 8, 12]))
        t0 = time.time()
        ret = mp.Manager().dict()
        try:
            mp.spawn(worker, args=(world, free_port(), par, ret), nprocs=world, join=True)
            ok = dict(ret) == {r: 1 for r in range(world)}
            why = "" if ok else f"ranks done: {dict(ret)}"
        except Exception as exc:  # noqa: BLE001
            ok, why = False, f"{type(exc).__name__}: {str(exc)[-600:]}"
        par.update(ok=ok, seconds=round(time.time() - t0, 1))
        if not ok:
            bad += 1
            par["why"] = why
        print("ok  " if ok else "FAIL", json.dumps(par), flush=True)
    print(f"{n} cases;", "SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
