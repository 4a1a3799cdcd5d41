"""Shared test scaffolding: small synthetic indexes in both the oracle's and the engine's form."""

from __future__ import annotations

import numpy as np

import oracle
from hipporag_amd import synth
from hipporag_amd.graph import bf16_bits_to_float


def make_case(num_vertices: int, num_edges: int, dim: int, seed: int, passage_frac: float = 0.125,
              power_law: bool = False):
    """Returns (kg, pass_bits, fact_bits, RefIndex) -- the same data for engine and oracle."""
    kg = synth.make_kg(num_vertices, num_edges, seed, passage_frac=passage_frac, power_law=power_law)
    pass_bits = synth.make_embeddings_np(kg.n_passages, dim, seed + 1)
    fact_bits = synth.make_embeddings_np(kg.n_facts, dim, seed + 2)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    p = oracle.column_normalize(a)
    index = oracle.RefIndex(fact_emb=bf16_bits_to_float(fact_bits), passage_emb=bf16_bits_to_float(pass_bits),
                            subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex,
                            num_chunks=kg.num_chunks, passage_vertex=kg.passage_vertex, p=p)
    return kg, pass_bits, fact_bits, index


from oracle.checks import (  # noqa: E402,F401  (the checkers live with the oracle: bench.py and smoke() use them too)
    ID_GAP_FLOOR, prior_noise_allowance, ranked_parity, tie_aware_equal, tie_aware_report, ulp4_report,
)


def write_test_report(name: str, record: dict) -> None:
    """Leave a small JSON record of a GPU test's parity figures under gpurun_out/test_reports/ (merged back from the
    GPU box) and on stdout (pytest -s); never fails the test."""
    import json
    import os
    print(f"[{name}] {json.dumps(record)}")
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = os.path.join(root, "gpurun_out", "test_reports")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(record, f)
    except OSError:
        pass
