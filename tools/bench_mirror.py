"""Mirror-level timing (GPU): `HippoRAG.retrieve()` of hipporag_amd/retriever.py end to end -- strings in, QuerySolution
lists out -- on a synthetic index, with the query encoder replaced by a table lookup (the embedding model is out of
scope).  Splits the wall clock into the device calls (engine) and the host work around them (the LLM-filter loop,
result materialisation: lists of `num_to_retrieve` document strings per query, as the reference returns them,
HippoRAG.py:501-507).

    python tools/bench_mirror.py [--config cfg2] [--queries 1024] [--num-to-retrieve 200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class TableEncoder:
    def __init__(self, table):
        self.table = table

    def batch_encode(self, texts, instruction=None, norm=True):
        kind = "f" if instruction and "fact" in instruction else "p"
        return np.stack([self.table[kind][t] for t in texts])


def main():
    import torch
    from hipporag_amd import synth
    from hipporag_amd.graph import bf16_bits_to_float
    from hipporag_amd.retriever import HippoRAG, RetrievalConfig
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3"])
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--num-to-retrieve", type=int, default=200)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    V, E, D = (100_000, 1_000_000, 768) if args.config == "cfg2" else (1_000_000, 10_000_000, 768)
    kg = synth.make_kg(V, E, 1236)
    pass_bits = synth.make_embeddings_np(kg.n_passages, D, 1)
    fact_bits = synth.make_embeddings_np(kg.n_facts, D, 2)
    qf_bits, _ = synth.make_queries_np(fact_bits, args.queries, seed=3)
    qp_bits, _ = synth.make_queries_np(pass_bits, args.queries, seed=4)
    queries = [f"query number {i}" for i in range(args.queries)]
    table = {"f": dict(zip(queries, bf16_bits_to_float(qf_bits))), "p": dict(zip(queries, bf16_bits_to_float(qp_bits)))}
    cfg = RetrievalConfig(embedding_precision="bf16", max_batch=args.batch)
    rag = HippoRAG.from_arrays(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                               kg.num_chunks, global_config=cfg, embedding_model=TableEncoder(table))
    rag.retrieve(queries[: args.batch], num_to_retrieve=args.num_to_retrieve)          # warm-up: engine, caches
    torch.cuda.synchronize()
    rag.rerank_time = rag.ppr_time = rag.all_retrieval_time = 0.0
    t0 = time.perf_counter()
    sols = rag.retrieve(queries, num_to_retrieve=args.num_to_retrieve)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert len(sols) == args.queries and len(sols[0].docs) == min(args.num_to_retrieve, kg.n_passages)
    print(json.dumps({"config": args.config, "queries": args.queries, "batch": args.batch,
                      "num_to_retrieve": args.num_to_retrieve, "wall_s": wall, "queries_per_s": args.queries / wall,
                      # the batches are pipelined (retriever.iter_batched_retrieve): these are the HOST's spans -- waiting for
                      # phase A + the filter loop; enqueueing phase B + waiting for its results; everything else
                      # (building the result lists), all three overlapped with the device working on the next batch
                      "host_phase_a_wait_plus_filter_loop_s": rag.rerank_time,
                      "host_phase_b_enqueue_plus_wait_s": rag.ppr_time,
                      "host_materialisation_s": wall - rag.rerank_time - rag.ppr_time}))


if __name__ == "__main__":
    main()
