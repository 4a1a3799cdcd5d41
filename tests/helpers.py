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


def tie_aware_equal(got_ids, ref_ids, ref_scores, rel_gap=4e-6, abs_gap=0.0):
    """Ranked ids equal, except inside runs of reference scores closer than the gap (tie classes,
    SURVEY.md 8c): there only set-equality is required; the class cut by the top-k boundary is
    not checked (its members may legitimately come from just beyond the boundary)."""
    got_ids = np.asarray(got_ids)
    ref_ids = np.asarray(ref_ids)
    if got_ids.shape != ref_ids.shape:
        return False
    if np.array_equal(got_ids, ref_ids):
        return True
    s = np.asarray(ref_scores, dtype=np.float64)
    n = len(ref_ids)
    start = 0
    while start < n:
        end = start + 1
        while end < n and abs(s[end - 1] - s[end]) <= max(abs_gap, rel_gap * abs(s[end - 1])):
            end += 1
        if end < n and set(got_ids[start:end].tolist()) != set(ref_ids[start:end].tolist()):
            return False
        start = end
    return True


def tie_aware_report(got_ids, ref_ids, ref_scores, rel_gap=4e-6, abs_gap=0.0):
    """tie_aware_equal plus HOW MUCH of the agreement is exact: {"equal": the tie-class-aware verdict,
    "exact_positions": ranks at which the two id lists agree outright, "n": ranks compared}.  The ranks that
    only agree as members of a tie class are n - exact_positions."""
    got = np.asarray(got_ids)
    ref = np.asarray(ref_ids)
    same = int((got == ref).sum()) if got.shape == ref.shape else 0
    return {"equal": bool(tie_aware_equal(got, ref, ref_scores, rel_gap=rel_gap, abs_gap=abs_gap)),
            "exact_positions": same, "n": int(ref.size)}
