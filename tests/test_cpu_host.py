"""CPU-only checks: the C-ABI library builds, loads and exports every symbol include/hrag.h
declares (no compute without a GPU); the product CSR builder agrees with the oracle's; the
bf16 helpers round like torch; the product path refuses to run without a GPU."""

import os
import re

import numpy as np
import pytest

import oracle
from hipporag_amd import _lib, synth
from hipporag_amd.graph import bf16_bits_to_float, build_csr, float_to_bf16_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "hrag.h")).read()
    declared = set(re.findall(r"\b(hrag_[a-z_0-9]+)\s*\(", header))
    declared -= {"hrag_status"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hrag_version() == _lib.HRAG_VERSION == 8            # 0 * 1000 + 8 (round 6: workspaces, stats, shard drivers; hrag_sim_topk_min_score)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The ctypes mirrors in hipporag_amd/_lib.py against include/hrag.h as gcc lays it out (sizes and
    the offset of every field): a drift here would corrupt hrag_engine_create's inputs silently."""
    import ctypes as C
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    structs = {"hrag_graph_desc": _lib.GraphDesc, "hrag_embed_desc": _lib.EmbedDesc,
               "hrag_fact_desc": _lib.FactDesc, "hrag_opts": _lib.Opts, "hrag_timings": _lib.Timings,
               "hrag_shard_layout": _lib.ShardLayout, "hrag_stats": _lib.Stats, "hrag_comm": _lib.Comm}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "hrag.h"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        cname, fname, val = ln.split()
        got[(cname, fname)] = int(val)
    for cname, cls in structs.items():
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_product_csr_builder_matches_oracle():
    kg = synth.make_kg(3000, 30000, seed=5, power_law=True)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    p = oracle.column_normalize(a)
    g = kg.csr
    assert g.nnz == 2 * 30000 == p.nnz
    np.testing.assert_array_equal(g.row_ptr, p.indptr)
    np.testing.assert_array_equal(g.col_idx, p.indices)
    np.testing.assert_allclose(g.val, p.data, rtol=1e-7)
    np.testing.assert_allclose(g.raw, a.data, rtol=1e-14)
    np.testing.assert_allclose(g.col_sum, np.asarray(a.sum(axis=0)).ravel(), rtol=1e-13)   # weighted degrees
    # column-stochastic (dangling columns aside)
    colsum = np.bincount(g.col_idx, weights=g.val.astype(np.float64), minlength=g.num_vertices)
    assert np.allclose(colsum[colsum > 0], 1.0, atol=1e-5)


def test_csr_builder_reference_edge_rules():
    # parallel edges (s,o)/(o,s) sum, self pairs dropped (HippoRAG.py:906-910, :1201, :1220)
    g = build_csr(4, [0, 1, 2, 0, 3], [1, 0, 2, 1, 0], [1.0, 1.0, 5.0, 0.5, 2.0])
    dense = np.zeros((4, 4))
    for i in range(4):
        for e in range(g.row_ptr[i], g.row_ptr[i + 1]):
            dense[i, g.col_idx[e]] = g.raw[e]
    want = np.zeros((4, 4))
    want[0, 1] = want[1, 0] = 2.5
    want[0, 3] = want[3, 0] = 2.0
    np.testing.assert_array_equal(dense, want)
    assert g.row_ptr[2] == g.row_ptr[3]          # vertex 2 only had a self loop -> dangling
    shard = g.rows(1, 3)
    assert shard.row_ptr[0] == 0 and shard.nnz == g.row_ptr[3] - g.row_ptr[1]


def test_bf16_rounding_matches_torch():
    import torch
    rng = np.random.default_rng(0)
    x = rng.standard_normal(10000).astype(np.float32)
    x[:4] = [0.0, -0.0, 1.0, 3.14159274]
    bits = float_to_bf16_bits(x)
    t = torch.from_numpy(x).to(torch.bfloat16)
    np.testing.assert_array_equal(bits.view(np.int16), t.view(torch.int16).numpy())
    np.testing.assert_array_equal(bf16_bits_to_float(bits), t.float().numpy())


def test_synthetic_kg_shape_contract():
    kg = synth.make_kg(8000, 80000, seed=9)
    assert kg.csr.nnz == 160000                   # nnz = 2E exactly
    assert kg.n_passages == 1000 and kg.n_entities == 7000 and kg.n_facts == 7000
    assert np.all(kg.passage_vertex == 7000 + np.arange(1000))
    deg = np.diff(kg.csr.row_ptr)
    assert deg[kg.passage_vertex].min() >= 1      # every passage links at least one entity
    assert np.all(kg.num_chunks[:7000] >= 1) and np.all(kg.num_chunks[7000:] == 0)
    assert kg.subj_vertex.max() < 7000 and np.all(kg.subj_vertex != kg.obj_vertex)
    kg2 = synth.make_kg(8000, 80000, seed=9)
    np.testing.assert_array_equal(kg.csr.col_idx, kg2.csr.col_idx)   # seeded => reproducible


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hipporag_amd.engine import HippoRAGEngine
    kg = synth.make_kg(200, 800, seed=1)
    emb = synth.make_embeddings_np(kg.n_passages, 64, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HippoRAGEngine(kg.csr, kg.passage_vertex, emb)
