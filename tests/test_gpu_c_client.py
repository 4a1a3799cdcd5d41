"""The drop-in boundary from the other side: a host written in PLAIN C (tests/c_client/hrag_client.c: include/hrag.h,
libhrag.so and the HIP runtime -- no Python, no torch in the process) stages an index, runs phase A / phase B under the
convergence contract, the same on a second workspace handle, and both phases through the one-call row-shard drivers at
world 1.  Compiled here with gcc (the header is C), run as its own process, and compared BIT FOR BIT with what the
Python wrapper returns for the same arrays -- the reference-side binding of INTEGRATION.md is exactly this surface
(HippoRAG.retrieve, src/hipporag/HippoRAG.py:413-499)."""

import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("batch", [9, 130])          # the fp16-state path / the staged e4m3 path + the shard drivers
def test_a_plain_c_host_gets_the_python_wrappers_results_bit_for_bit(tmp_path, batch):
    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine
    from tests.helpers import make_case
    if not shutil.which("gcc") or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    exe = tmp_path / "hrag_client"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c_client", "hrag_client.c"),
                           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-L", os.path.join(ROOT, "hipporag_amd"), "-lhrag", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "hipporag_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    kg, pass_bits, fact_bits, _ = make_case(5000, 50000, 64, seed=9091)
    k, kf, b = 40, 5, batch
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=1)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=2)
    g = kg.csr
    case = tmp_path / "case.bin"
    with open(case, "wb") as f:
        np.array([g.num_vertices, g.nnz, kg.n_passages, kg.n_facts, 64, b, k, kf], dtype=np.int64).tofile(f)
        for arr, dt in ((g.row_ptr, np.int32), (g.col_idx, np.int32), (g.val, np.float32), (g.col_sum, np.float64),
                        (kg.passage_vertex, np.int32), (pass_bits, np.uint16), (fact_bits, np.uint16),
                        (kg.subj_vertex, np.int32), (kg.obj_vertex, np.int32), (kg.num_chunks, np.int32),
                        (qf_bits, np.uint16), (qp_bits, np.uint16)):
            np.ascontiguousarray(arr, dtype=dt).tofile(f)
    out = tmp_path / "out.bin"
    p = subprocess.run([str(exe), str(case), str(out)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "hrag_client OK" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    raw = np.fromfile(out, dtype=np.uint8)
    pos = 0

    def take(shape, dt):
        nonlocal pos
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        a = raw[pos: pos + n].view(dt).reshape(shape)
        pos += n
        return a

    c_fidx, c_fsc = take((b, kf), np.int32), take((b, kf), np.float32)
    c_didx, c_dsc, c_flags = take((b, k), np.int32), take((b, k), np.float32), take((b,), np.int32)
    c_res, c_used = take((b,), np.float32), take((b,), np.int32)
    w_didx, w_dsc = take((b, k), np.int32), take((b, k), np.float32)
    # ---- the Python wrapper on the same arrays
    dev = torch.device("cuda", 0)

    def bf16(bits):
        return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(dev).view(torch.bfloat16)

    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=b, max_topk=k) as eng:
        idx, sc = eng.score_facts(bf16(qf_bits), k=kf)
        cnt = torch.full((b,), kf, dtype=torch.int32, device=dev)
        o = eng.retrieve(bf16(qp_bits), idx, sc, cnt, ppr_iters=20, k=k, ppr_tol=1.5e-6, ppr_max_iters=29)
        torch.cuda.synchronize()
        want = [t.cpu().numpy() for t in (idx, sc, o.doc_idx, o.doc_score, o.flags, o.residual, o.iters_used)]
    for got, w in zip((c_fidx, c_fsc, c_didx, c_dsc, c_flags, c_res, c_used), want):
        np.testing.assert_array_equal(got, w)
    np.testing.assert_array_equal(w_didx, want[2])            # the workspace handle: the same answer
    np.testing.assert_array_equal(w_dsc, want[3])
    assert np.all(c_flags == 0) and c_used.min() >= 20
    if b > 64:                                                # hrag_shard_score_facts_all / hrag_shard_retrieve, world 1
        s_fidx, s_fsc = take((b, kf), np.int32), take((b, kf), np.float32)
        s_didx, s_dsc = take((b, k), np.int32), take((b, k), np.float32)
        np.testing.assert_array_equal(s_fidx, want[0])
        np.testing.assert_array_equal(s_fsc, want[1])
        np.testing.assert_array_equal(s_didx, want[2])
        np.testing.assert_array_equal(s_dsc, want[3])
    assert pos == raw.size
