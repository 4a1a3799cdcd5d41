"""HRAG_OPT_ACCEL: the stages of the fp8-state PPR as Chebyshev steps on the real spectrum of the sweep operator
(csrc/shard.hip ppr8_plan_accel).  `ppr_iters` then names an ACCURACY -- that of so many plain sweeps of the iteration
igraph's PRPACK solve stands for (reference src/hipporag/HippoRAG.py:1736-1743) -- and fewer sweeps run.  Off by
default; never the benchmark headline.  The adversarial graphs under the convergence contract WITH the flag:
tests/test_gpu_fp8_adversarial.py::test_accelerated_stages_on_adversarial_graphs_under_the_contract."""
import math

import numpy as np
import pytest

import oracle
from hipporag_amd import synth
from hipporag_amd._lib import PPR_ERR_FLOOR_F16, PPR_ERR_FLOOR_FP8, PPR_ERR_K, PPR_TOL_MIN
from hipporag_amd.graph import bf16_bits_to_float
from tests.helpers import prior_noise_allowance, write_test_report

pytestmark = pytest.mark.gpu


def _t(x, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def _bf16(bits, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(device).view(torch.bfloat16)


def accel_sweeps(iters, damping, measured=False):
    """Sweeps of the plan 1, 3, 3, ... (+ 1 plain closing sweep under the contract) that stands for `iters` plain
    sweeps (csrc/shard.hip ppr8_plan_accel); `iters` itself where the variant saves nothing."""
    x = 1.0 / damping
    k3 = max(1.0 / (4 * x ** 3 - 3 * x), 1.0 / 14.0)
    n3 = math.ceil((iters - 1) * math.log(1.0 / damping) / math.log(1.0 / k3) - 1e-9)
    total = 1 + 3 * n3 + (1 if measured else 0)
    return total if (damping >= 0.2 and n3 >= 2 and total < iters) else iters


def accel16_sweeps(iters, damping, margin=16.0):
    """Sweeps of the accelerated two-stage fp16 plan (csrc/engine.hip accel_plan16, restated): the smallest
    K1 + 1 + K2 + 2 whose Chebyshev bound reaches damping^iters / margin; `iters` where that saves nothing."""
    if not (0.2 <= damping <= 0.62) or iters < 16:
        return iters
    t = lambda k: min(math.cosh(k * math.acosh(1.0 / damping)), 2048.0)
    target = margin * damping ** -iters
    return min([k1 + k2 + 3 for k1 in range(3, 15) for k2 in range(2, 15)
                if t(k1) * t(k2 + 1) / damping ** 2 >= target] + [iters])


def test_accel16_plan_restatement():
    assert accel16_sweeps(20, 0.5) == 16 and accel16_sweeps(20, 0.5, margin=4.0) == 14
    assert accel16_sweeps(28, 0.6) == 28 and accel16_sweeps(20, 0.6) == 15 and accel16_sweeps(16, 0.3) == 16


@pytest.mark.parametrize("b,power_law", [(130, False), (257, True)])
def test_sixteen_accelerated_sweeps_stand_for_twenty_plain_ones(gpu_device, b, power_law):
    import torch
    from hipporag_amd._lib import OPT_ACCEL
    from hipporag_amd.engine import HippoRAGEngine
    from tests.helpers import make_case
    kg, pass_bits, fact_bits, index = make_case(16000, 160000, 64, seed=41 + b, power_law=power_law)
    n_p = kg.n_passages
    assert n_p <= 2048 and accel_sweeps(20, 0.5) == 16 and accel_sweeps(20, 0.5, True) == 17
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    got = {}
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        for acc in (False, True, False):                   # and back: the flag is a runtime switch
            eng.set_flags(OPT_ACCEL, acc)
            out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p)
            torch.cuda.synchronize()
            assert eng.timings()["slab_width"] == 128
            used, flags = out.iters_used.cpu().numpy(), out.flags.cpu().numpy()
            assert np.all(flags == 0), np.unique(flags)    # in particular no value left the e4m3 range
            assert np.all(used == (16 if acc else 20))
            res = (out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy(), out.residual.cpu().numpy())
            if acc in got:                                 # the second plain run reproduces the first bit for bit
                assert all(np.array_equal(x, y) for x, y in zip(got[acc], res))
            got[acc] = res
        # under the contract the plan ends on a plain sweep: 17 sweeps, and the measure means what it means without the flag
        eng.set_flags(OPT_ACCEL, True)
        con = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p, ppr_tol=1.5e-6, ppr_max_iters=30)
        torch.cuda.synchronize()
        con_used, con_res = con.iters_used.cpu().numpy(), con.residual.cpu().numpy()
        con_ids, con_sc, con_flags = con.doc_idx.cpu().numpy(), con.doc_score.cpu().numpy(), con.flags.cpu().numpy()
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    worst = {False: 0.0, True: 0.0, "contract": 0.0}
    under = 0.0
    for q in range(0, b, max(1, b // 10)):
        want = oracle.retrieve_one(index, qf[q], qp[q]).x[index.passage_vertex]
        allow = prior_noise_allowance(index, qp[q])
        nz = want > 0
        for key, (ids, scs) in (("contract", (con_ids, con_sc)), (False, got[False][:2]), (True, got[True][:2])):
            full = np.empty(n_p)
            full[ids[q]] = scs[q]
            e = float((np.abs(full[nz] / want[nz] - 1) - allow[nz]).max())
            worst[key] = max(worst[key], e)
            if key == "contract":
                under = max(under, e / max(float(con_res[q]), 1e-30))
                # the bound include/hrag.h states for a met tolerance, query by query
                assert e <= max(PPR_ERR_K * float(con_res[q]), PPR_ERR_FLOOR_FP8), (q, e, float(con_res[q]))
            else:       # a fixed count reports what it leaves: the same inequality holds for the reported residual
                assert e <= max(PPR_ERR_K * float(got[key][2][q]), PPR_ERR_FLOOR_FP8), (key, q, e, float(got[key][2][q]))
    write_test_report(f"accel_small_graph_b{b}", {"worst_rel_err_plain20": worst[False], "worst_rel_err_accel16": worst[True],
                                                   "worst_rel_err_accel_contract": worst["contract"],
                                                   "sweeps_contract_min_max": [int(con_used.min()), int(con_used.max())],
                                                   "max_true_error_over_reported_residual": under})
    assert worst[False] < 1e-5, worst
    # tolerance 0: the accuracy `ppr_iters` names, up to the factor include/hrag.h states (3x the truncation error)
    assert worst[True] < 3 * max(worst[False], 2e-6), worst
    # the contract: at least the 17 sweeps of the measured plan, nothing flagged, every score inside the bar with margin
    assert con_used.min() >= 17 and con_used.max() <= 30 and np.all(con_flags == 0)
    assert worst["contract"] < 1e-5 / 1.5, worst
    assert np.all(con_res <= 1.5e-6)


def test_accelerated_valid_inputs_never_saturate_and_small_damping_keeps_the_plain_plan(gpu_device):
    import torch
    from hipporag_amd._lib import OPT_ACCEL
    from hipporag_amd.engine import HippoRAGEngine
    from tests.test_gpu_fp8_adversarial import _small_engine_inputs
    b = 70
    kg, pass_bits, fact_bits, qf, qp = _small_engine_inputs(b, gpu_device)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=b, max_topk=50, flags=OPT_ACCEL) as eng:
        idx, sc = eng.score_facts(qf, k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        for pw in (1e-6, 0.05, 1e4):
            out = eng.retrieve(qp, idx, sc, cnt, passage_node_weight=pw, ppr_iters=20, k=50)
            torch.cuda.synchronize()
            assert eng.timings()["slab_width"] == 128
            assert np.all(out.flags.cpu().numpy() == 0), pw
            assert int(out.iters_used.max()) == 16
        for damping, iters in ((0.3, 16), (0.6, 28)):      # (0.7 needs 39 > 30 sweeps: not an fp8-state batch at all)
            out = eng.retrieve(qp, idx, sc, cnt, damping=damping, ppr_iters=iters, k=50)
            torch.cuda.synchronize()
            assert eng.timings()["slab_width"] == 128 and np.all(out.flags.cpu().numpy() == 0), damping
            assert int(out.iters_used.min()) == int(out.iters_used.max()) == accel_sweeps(iters, damping), damping
    assert accel_sweeps(16, 0.3) == 16 and accel_sweeps(28, 0.6) == 19


@pytest.mark.parametrize("b", [1, 4, 8, 40, 64])
def test_accelerated_fp16_states_sixteen_sweeps_stand_for_twenty(gpu_device, b):
    """HRAG_OPT_ACCEL on the two-stage fp16 states (B <= 8: ppr_sv.hip; 9 .. 64: ppr16.hip): Chebyshev steps in both
    stages, then a plain correction sweep and the plain final sweep (csrc/engine.hip accel_plan16): 16 sweeps for the
    accuracy of 20 plain ones at damping 0.5 (14 until the round-6 soaks: the margin of the plan went from 4 to 16, see
    accel16_sweeps below) -- with ppr_tol = 0 only: under a tolerance these states keep the plain
    plan + its device-side extension (csrc/engine.hip accel_plan16 on why).  Same bars as the plain path; the flag is a
    runtime switch; the measure still reads a plain sweep's update."""
    import torch
    from hipporag_amd._lib import OPT_ACCEL
    from hipporag_amd.engine import HippoRAGEngine
    from tests.helpers import make_case
    kg, pass_bits, fact_bits, index = make_case(16000, 160000, 64, seed=77, power_law=(b % 2 == 0))
    n_p = kg.n_passages
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    got = {}
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        for acc in (False, True, False):
            eng.set_flags(OPT_ACCEL, acc)
            out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p)
            torch.cuda.synchronize()
            assert eng.timings()["slab_width"] != 128      # an fp16 state served the call
            used, flags = out.iters_used.cpu().numpy(), out.flags.cpu().numpy()
            assert np.all(flags == 0), np.unique(flags)
            assert np.all(used == (accel16_sweeps(20, 0.5) if acc else 20)), used
            res = (out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy(), out.residual.cpu().numpy())
            if acc in got:
                assert all(np.array_equal(x, y) for x, y in zip(got[acc], res))
            got[acc] = res
        eng.set_flags(OPT_ACCEL, True)
        con = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p, ppr_tol=1.5e-6, ppr_max_iters=30)
        torch.cuda.synchronize()
        tol_used, con_res, con_flags = int(con.iters_used.max()), float(con.residual.max()), con.flags.cpu().numpy()
        got["contract"] = (con.doc_idx.cpu().numpy(), con.doc_score.cpu().numpy(), con.residual.cpu().numpy())
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    worst = {False: 0.0, True: 0.0, "contract": 0.0}
    under = 0.0
    for q in range(0, b, max(1, b // 8)):
        want = oracle.retrieve_one(index, qf[q], qp[q]).x[index.passage_vertex]
        allow = prior_noise_allowance(index, qp[q])
        nz = want > 0
        for acc, (ids, scs, res) in got.items():
            full = np.empty(n_p)
            full[ids[q]] = scs[q]
            e = float((np.abs(full[nz] / want[nz] - 1) - allow[nz]).max())
            worst[acc] = max(worst[acc], e)
            if acc is True:
                under = max(under, e / max(float(res[q]), 1e-30))
            assert e <= max(PPR_ERR_K * float(res[q]), PPR_ERR_FLOOR_F16), (acc, q, e, float(res[q]))   # include/hrag.h's bound
    write_test_report(f"accel_fp16_state_b{b}", {"worst_rel_err_plain20": worst[False], "worst_rel_err_accel14": worst[True],
                                                  "worst_rel_err_accel_contract": worst["contract"], "sweeps_accel_contract": tol_used,
                                                  "residual_max_plain20": float(got[False][2].max()),
                                                  "residual_max_accel14": float(got[True][2].max()),
                                                  "max_true_error_over_reported_residual_accel": under})
    assert worst[False] < 1e-5 and worst[True] < 1e-5, worst
    assert worst[True] < 3e-6, worst                           # well inside the bar (the plain plan beats its bound here)
    assert float(got[True][2].max()) < 2e-5                    # the measure stays a plain sweep's update: no 60x over-read
    # under a tolerance these states keep the PLAIN plan even with the flag on (the fp16 rounding of the larger correction
    # an accelerated first stage leaves puts a floor of 3e-6 .. 1e-5 under the measured residual): 20 sweeps, nothing flagged
    assert tol_used == 20 and con_res <= 1.5e-6 and np.all(con_flags == 0), (tol_used, con_res)
    assert worst["contract"] < 1e-5 / 1.5, worst


def test_a_tolerance_below_the_arithmetic_floor_is_rejected(gpu_device):
    """include/hrag.h: error <= max(HRAG_PPR_ERR_K * residual, floor) -- a ppr_tol below HRAG_PPR_TOL_MIN promises nothing
    fp32 arithmetic can deliver (the measure of a converged iterate reads 1e-8 next to a true error of 2e-7), so
    hrag_retrieve and the shard entry point refuse it instead of letting a caller believe it; 0 (fixed count) and
    HRAG_PPR_TOL_MIN itself are accepted."""
    import torch
    from hipporag_amd._lib import HragError
    from hipporag_amd.engine import HippoRAGEngine
    from tests.test_gpu_fp8_adversarial import _small_engine_inputs
    for b in (8, 40, 130):
        kg, pass_bits, fact_bits, qf, qp = _small_engine_inputs(b, gpu_device)
        with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                            kg.num_chunks, max_batch=b, max_topk=50) as eng:
            idx, sc = eng.score_facts(qf, k=5)
            cnt = _t(np.full(b, 5, np.int32), gpu_device)
            with pytest.raises(HragError, match="HRAG_PPR_TOL_MIN"):
                eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50, ppr_tol=1e-9, ppr_max_iters=30)
            ok = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50, ppr_tol=PPR_TOL_MIN, ppr_max_iters=30)
            fixed = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50)
            torch.cuda.synchronize()
            assert int(fixed.iters_used.max()) == 20 and int(ok.iters_used.max()) >= 20
