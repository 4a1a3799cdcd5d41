#!/usr/bin/env python
"""What hrag_workspace_create buys a server: aggregate queries/s of T threads, each with its own workspace and stream on
ONE index, against one thread -- for the batch sizes an interactive service sees (B = 1, 8, 64) and the benchmark's 256.

    python tools/bench_workspaces.py [--config cfg3] [--threads 1,2,4] [--batches 1,8,64,256]

Every thread runs phase A + identity filter + phase B on its own fresh queries `rounds` times; the wall time of the slowest
thread counts.  Results of thread 0 are checked against the single-handle call (bit-identical: tests/test_gpu_workspace.py
carries the full check)."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CONFIGS  # noqa: E402
from hipporag_amd import synth  # noqa: E402
from hipporag_amd.engine import HippoRAGEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--threads", default="1,2,4")
    ap.add_argument("--batches", default="1,8,64,256")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_workspaces.json"))
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    V, E, D, seed = cfg["V"], cfg["E"], cfg["D"], cfg["seed"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    kg = synth.make_kg(V, E, seed)
    pemb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, dev)
    femb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, dev)
    res = {}
    for B in [int(b) for b in args.batches.split(",")]:
        eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                             max_batch=B, max_topk=200)
        st = eng.stats()
        row = {"index_bytes": st["index_bytes"], "workspace_bytes": st["workspace_bytes"]}
        for T in [int(t) for t in args.threads.split(",")]:
            handles = [eng] + [eng.workspace() for _ in range(T - 1)]
            qs = [(synth.make_queries_torch(femb, B, 100 + t)[0], synth.make_queries_torch(pemb, B, 500 + t)[0]) for t in range(T)]
            cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
            start, times, errors = threading.Barrier(T), [0.0] * T, []

            def worker(t):
                try:
                    torch.cuda.set_device(dev)
                    h, (qf, qp) = handles[t], qs[t]
                    stream = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(stream):
                        for _ in range(3):
                            i, s = h.score_facts(qf, k=5)
                            h.retrieve(qp, i, s, cnt, ppr_iters=20, k=200)
                        stream.synchronize()
                        start.wait()
                        t0 = time.perf_counter()
                        for _ in range(args.rounds):
                            i, s = h.score_facts(qf, k=5)
                            h.retrieve(qp, i, s, cnt, ppr_iters=20, k=200)
                        stream.synchronize()
                        times[t] = time.perf_counter() - t0
                except Exception as exc:  # noqa: BLE001
                    errors.append(repr(exc))
                    try:
                        start.abort()
                    except Exception:
                        pass

            ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            if errors:
                row[f"threads{T}"] = {"error": errors[:2]}
            else:
                wall = max(times)
                row[f"threads{T}"] = {"queries_per_s": T * B * args.rounds / wall, "ms_per_call_per_thread": wall * 1e3 / args.rounds}
            for h in handles[1:]:
                h.close()
        base = row.get("threads1", {}).get("queries_per_s")
        for k, v in row.items():
            if k.startswith("threads") and base and "queries_per_s" in v:
                v["x_one_thread"] = v["queries_per_s"] / base
        res[B] = row
        print(B, json.dumps(row), flush=True)
        eng.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
