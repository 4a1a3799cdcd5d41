"""The staged fp8 PPR state (csrc/ppr8.hip) through hrag_retrieve on graphs chosen to break it: slow
mixing (ring), hub rows / bipartite structure (star forest), two clusters joined by one weak edge with
edge weights spanning 1e-3 .. 1e3, a tiny component that holds all the seeds of some queries, and a
larger damping factor.  Every case compares ALL passage scores (k = Np) with the oracle and requires
that no value had to be clamped to the e4m3 range (flags bit 3, HRAG_FLAG_FP8_SATURATED).

Bars.  The parity bar is 1e-5 relative against the exact solution (PRPACK solves to 1e-10,
HippoRAG.py:1736-1743) and EVERY case asserts it with a margin of 1.5 (worst < 6.7e-6) under the default tolerance.  A fixed number of sweeps cannot meet it on a slowly mixing
graph whatever the state type (the fp64 20-sweep iterate itself is 7e-4 off on the ring, 1.9e-6 on the star
forest): what meets it is the convergence contract of hrag_retrieve (include/hrag.h) -- ppr_tol on the measured
relative update of the passage scores, device-side extension stages on the fp8 state (star forest: 26 sweeps),
HRAG_FLAG_NOT_CONVERGED where 30 sweeps do not suffice (ring) and the repeat of exactly those queries on the
fp32 state with the sweeps their residual asks for (HippoRAGEngine.retrieve_converged)."""

import numpy as np
import pytest

import oracle
from hipporag_amd import synth
from hipporag_amd.graph import bf16_bits_to_float, build_csr
from tests.helpers import prior_noise_allowance, write_test_report

pytestmark = pytest.mark.gpu


def _t(x, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def _bf16(bits, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(device).view(torch.bfloat16)


def _index(n, src, dst, w, pv, dim, seed, pinned_facts=()):
    """Engine arrays + oracle index for an arbitrary weighted edge list; facts connect random non-passage
    vertices (pinned_facts: (subj, obj) pairs placed first)."""
    rng = np.random.default_rng(seed)
    csr = build_csr(n, src, dst, w)
    is_p = np.zeros(n, bool)
    is_p[pv] = True
    ents = np.flatnonzero(~is_p)
    n_f = max(64, len(ents) // 2)
    subj = ents[rng.integers(0, len(ents), n_f)]
    obj = ents[rng.integers(0, len(ents), n_f)]
    obj = np.where(obj == subj, ents[(np.searchsorted(ents, subj) + 1) % len(ents)], obj)
    for i, (s, o) in enumerate(pinned_facts):
        subj[i], obj[i] = s, o
    num_chunks = np.zeros(n, np.int32)
    num_chunks[ents] = rng.integers(1, 4, len(ents))
    pass_bits = synth.make_embeddings_np(len(pv), dim, seed + 1)
    fact_bits = synth.make_embeddings_np(n_f, dim, seed + 2)
    a = oracle.build_symmetric_csr(n, src, dst, w)
    index = oracle.RefIndex(fact_emb=bf16_bits_to_float(fact_bits), passage_emb=bf16_bits_to_float(pass_bits),
                            subj_vertex=subj.astype(np.int32), obj_vertex=obj.astype(np.int32), num_chunks=num_chunks,
                            passage_vertex=np.asarray(pv, np.int32), p=oracle.column_normalize(a))
    return csr, pass_bits, fact_bits, index


def _ring():
    n = 4000
    i = np.arange(n)
    return n, i, (i + 1) % n, np.ones(n), i[::8], ()


def _stars():
    n = 5000
    hubs, leaves = np.arange(10), np.arange(10, n)
    src = np.concatenate([leaves, hubs[:-1]])
    dst = np.concatenate([hubs[(leaves - 10) % 10], hubs[1:]])
    return n, src, dst, np.ones(len(src)), leaves[::8], ()


def _barbell():
    rng = np.random.default_rng(7)
    n = 2000
    s1, d1 = rng.integers(0, 1000, 20000), rng.integers(0, 1000, 20000)
    s2, d2 = rng.integers(1000, 2000, 20000), rng.integers(1000, 2000, 20000)
    src, dst = np.concatenate([s1, s2, [0]]), np.concatenate([d1, d2, [1999]])
    w = np.concatenate([10.0 ** rng.uniform(-3, 3, 40000), [1e-3]])
    keep = src != dst
    return n, src[keep], dst[keep], w[keep], np.arange(0, n, 8), ()


def _tiny_component():
    """3000-vertex random graph + a 12-vertex clique attached by ONE edge of weight 1e-3; the first facts
    live inside the clique, so the queries aimed at them seed only the tiny component."""
    rng = np.random.default_rng(3)
    n, nb = 3012, 12
    s, d = rng.integers(0, 3000, 30000), rng.integers(0, 3000, 30000)
    cs, cd = np.triu_indices(nb, 1)
    src = np.concatenate([s, 3000 + cs, [17]])
    dst = np.concatenate([d, 3000 + cd, [3000]])
    w = np.concatenate([rng.uniform(0.5, 3.0, 30000), np.ones(len(cs)), [1e-3]])
    keep = src != dst
    pv = np.concatenate([np.arange(0, 3000, 6), [3003, 3007]])          # two passages inside the clique
    pinned = [(3001, 3002), (3004, 3005), (3006, 3008), (3009, 3010)]
    return n, src[keep], dst[keep], w[keep], pv, pinned


def _sparse_power_law():
    """The generator's power-law graph at mean degree 6 with 30 % passages (what tools/soak_random.py tripped over): the
    max-norm of the residual contracts visibly slower than damping^m per stage once the e4m3 rounding is in it."""
    kg = synth.make_kg(12000, 36000, 31506654, passage_frac=0.3, power_law=True)
    return kg.num_vertices, kg.src, kg.dst, kg.weight, np.asarray(kg.passage_vertex)[:2000], ()


CASES = {"ring": (_ring, 0.5, 20), "stars": (_stars, 0.5, 20), "barbell_wild_weights": (_barbell, 0.5, 20),
         "tiny_component": (_tiny_component, 0.5, 20), "barbell_damping_0.6": (_barbell, 0.6, 28),
         "tiny_component_24_sweeps": (_tiny_component, 0.5, 24),
         # small damping: damping^m per stage is BELOW what the e4m3 rounding of a stage puts back, the static scale chain
         # drifted out of the range there (round 4, found by the randomised soak); the plan measures its scales instead
         "ring_damping_0.3": (_ring, 0.3, 16), "stars_damping_0.3": (_stars, 0.3, 16),
         "sparse_power_law_damping_0.3": (_sparse_power_law, 0.3, 16), "sparse_power_law_damping_0.4": (_sparse_power_law, 0.4, 16),
         "sparse_power_law": (_sparse_power_law, 0.5, 20)}


@pytest.mark.parametrize("b", [65, 256])
@pytest.mark.parametrize("name", sorted(CASES))
def test_fp8_state_on_adversarial_graphs_all_passages(gpu_device, name, b):
    import dataclasses
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    make, damping, iters = CASES[name]
    n, src, dst, w, pv, pinned = make()
    csr, pass_bits, fact_bits, index = _index(n, src, dst, w, pv, 64, seed=11, pinned_facts=pinned)
    index = dataclasses.replace(index, damping=damping)
    n_p = len(pv)
    assert n_p <= 2048
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    for i in range(len(pinned)):
        qf_bits[i] = fact_bits[i]                        # these queries' best fact is a pinned one
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    from hipporag_amd._lib import FLAG_NOT_CONVERGED
    tol = 1.5e-6                                          # RetrievalConfig.ppr_tol: worst-case under-reading of the measure
                                                          # (0.29) still lands at 5.2e-6, the bar with a factor 1.9 to spare
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        # (1) the engine call by itself: the fp8 state, extension stages decided on the device, a flag where its
        # 30 sweeps do not reach the tolerance
        raw = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, damping=damping, ppr_iters=iters, k=n_p,
                           ppr_tol=tol, ppr_max_iters=400)
        torch.cuda.synchronize()
        assert eng.timings()["slab_width"] == 128        # the staged fp8 state served the call
        raw_flags, raw_used = raw.flags.cpu().numpy(), raw.iters_used.cpu().numpy()
        raw_resid = raw.residual.cpu().numpy()
        # (2) the contract end to end: flagged queries repeated on the fp32 state
        out = eng.retrieve_converged(_bf16(qp_bits, gpu_device), idx, sc, cnt, damping=damping, ppr_iters=iters,
                                     k=n_p, ppr_tol=tol, ppr_max_iters=400)
        torch.cuda.synchronize()
        got_idx, got_sc, flags = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy(), out.flags.cpu().numpy()
        resid, used = out.residual.cpu().numpy(), out.iters_used.cpu().numpy()
    assert np.all(raw_flags & ~FLAG_NOT_CONVERGED == 0), np.unique(raw_flags)   # in particular no HRAG_FLAG_FP8_SATURATED
    assert np.all(flags == 0), np.unique(flags)           # nothing is left unconverged
    assert np.all(resid <= tol) and np.all(resid >= 0)
    assert np.all((raw_resid > tol) == ((raw_flags & FLAG_NOT_CONVERGED) != 0))
    assert raw_used.min() >= iters and raw_used.max() <= 30
    if name == "ring":                                    # 30 sweeps are not enough: flagged, then repeated with more
        assert (raw_flags & FLAG_NOT_CONVERGED).any() and used.max() > 30
    elif name == "stars":                                 # the device added stages, nothing needed repeating
        assert raw_used.max() > iters and not (raw_flags & FLAG_NOT_CONVERGED).any()
    elif "barbell" in name or "tiny" in name:             # well-mixing graphs: at most one short extension stage, no flag
        assert raw_used.max() <= iters + 3 and np.all(raw_flags == 0)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    check = list(range(len(pinned))) + list(range(len(pinned), b, max(1, b // 12)))
    worst, bound_k = 0.0, 0.0
    from hipporag_amd._lib import PPR_ERR_FLOOR_FP8, PPR_ERR_K
    from tests.helpers import write_test_report
    for q in check:
        exact = oracle.retrieve_one(index, qf[q], qp[q])
        want = exact.x[index.passage_vertex]
        full = np.empty(n_p)
        full[got_idx[q]] = got_sc[q]                      # k = Np: every passage's score came back
        assert np.array_equal(np.sort(got_idx[q]), np.arange(n_p)), q
        nz = want > 0
        # beyond what the reference itself leaves undefined (fp32 dot noise in the prior of near-minimum passages)
        allow = prior_noise_allowance(index, qp[q])
        e = float((np.abs(full[nz] / want[nz] - 1) - allow[nz]).max())
        worst = max(worst, e)
        # the bound include/hrag.h states for a met tolerance, query by query: the floor of the state that produced the
        # final answer (a repeated query ends on the fp32 state: the smaller floor; this side takes the larger one)
        bound_k = max(bound_k, e / max(float(resid[q]), 1e-30) if e > PPR_ERR_FLOOR_FP8 else 0.0)
        assert e <= max(PPR_ERR_K * float(resid[q]), PPR_ERR_FLOOR_FP8), (name, b, q, e, float(resid[q]))
        assert np.all(full[~nz] == 0), q
    write_test_report(f"adversarial_{name}_b{b}", {"worst_rel_err": worst, "residual_max": float(resid.max()),
                                                   "sweeps_max": int(used.max()), "error_over_residual_above_the_floor": bound_k})
    assert worst < 1e-5 / 1.5, (name, b, worst)           # the parity bar WITH a margin of 1.5, every case


def test_fixed_sweep_count_reports_the_residual_it_leaves(gpu_device):
    """ppr_tol = 0 is BASELINE.json's fixed 20 sweeps: nothing is extended or flagged, and the residual the call
    reports says what that costs -- on the ring it reads > 1e-3 (the true error of those scores is 7e-4 .. 2e-3),
    on the benchmark generator < 3e-6.  The measure never falls below 1 / 4 of the true error where that error
    is above the fp32 noise floor."""
    import dataclasses
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    n, src, dst, w, pv, pinned = _ring()
    csr, pass_bits, fact_bits, index = _index(n, src, dst, w, pv, 64, seed=11)
    b, n_p = 65, len(pv)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, _t(np.full(b, 5, np.int32), gpu_device), ppr_iters=20, k=n_p)
        torch.cuda.synchronize()
        got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
        resid, used, flags = out.residual.cpu().numpy(), out.iters_used.cpu().numpy(), out.flags.cpu().numpy()
    assert np.all(used == 20) and np.all(flags == 0)
    for q in range(0, b, 6):
        want = oracle.retrieve_one(index, qf[q], qp[q]).x[index.passage_vertex]
        err = float(np.abs(got_sc[q] / want[got_idx[q]] - 1).max())
        assert err > 1e-5 and resid[q] > 0.25 * err, (q, err, resid[q])    # 20 sweeps are NOT enough here, and it says so
    kg, pb, fb, qfb, qpb = _small_engine_inputs(70, gpu_device)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pb, fb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=70, max_topk=50) as eng:
        idx, sc = eng.score_facts(qfb, k=5)
        out = eng.retrieve(qpb, idx, sc, _t(np.full(70, 5, np.int32), gpu_device), ppr_iters=20, k=50)
        torch.cuda.synchronize()
        assert float(out.residual.max()) < 3e-6 and int(out.iters_used.max()) == 20


def _small_engine_inputs(b, device):
    kg = synth.make_kg(4000, 40000, 9)
    pass_bits = synth.make_embeddings_np(kg.n_passages, 64, 1)
    fact_bits = synth.make_embeddings_np(kg.n_facts, 64, 2)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=1)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=2)
    return kg, pass_bits, fact_bits, _bf16(qf_bits, device), _bf16(qp_bits, device)


def test_valid_inputs_with_extreme_prior_to_seed_ratios_never_saturate(gpu_device):
    """v is rescaled per query so that max v/d lands in (1/2, 1], whatever the ratio of the passage prior to
    the seed weights: no valid input may raise HRAG_FLAG_FP8_SATURATED."""
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    b = 70
    kg, pass_bits, fact_bits, qf, qp = _small_engine_inputs(b, gpu_device)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=b, max_topk=50) as eng:
        idx, sc = eng.score_facts(qf, k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        for pw in (1e-6, 0.05, 1e4):
            out = eng.retrieve(qp, idx, sc, cnt, passage_node_weight=pw, ppr_iters=20, k=50)
            torch.cuda.synchronize()
            assert eng.timings()["slab_width"] == 128
            assert np.all(out.flags.cpu().numpy() == 0), pw


def test_a_violated_scale_bound_raises_the_saturation_flag(gpu_device):
    """The static fp8 scales rest on max v/d <= 1 after the per-query rescaling.  Through the row-shard
    entry points the caller supplies the (all-reduced) maximum; handing in one that is 64x too small is
    exactly a violated bound: the result must carry flags bit 3 for the affected queries instead of
    silently clipped scores -- and the correct maximum must not."""
    import torch
    from hipporag_amd import dist as hd
    from hipporag_amd.engine import ShardStages
    from hipporag_amd._lib import FLAG_FP8_SATURATED
    b = 96
    kg, pass_bits, fact_bits, qf, qp = _small_engine_inputs(b, gpu_device)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, 1, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    eng = hd.build_shard_engine(sidx, pass_bits, fact_bits, 0, max_batch=b, max_topk=50)
    st = ShardStages(eng)
    lay = st.shard_layout(b, 1)
    bufs = [st.new_state(lay) for _ in range(3)]
    idx, val, mn_f, mx_f = st.shard_score_facts(qf, 5)
    sc = (val - mn_f[:, None]) / (mx_f - mn_f)[:, None]
    cnt = _t(np.full(b, 5, np.int32), gpu_device)
    got = {}
    for label, shrink in (("correct", 1.0), ("violated", 1.0 / 64)):
        mn, mx = st.shard_passage_scores(qp)
        sv, sw, scnt, flags = st.seeds(idx, sc.contiguous(), cnt, 5)
        zmax, mass = st.shard_prior_stats(mn, mx, 0.05, flags)
        zmax = (zmax * shrink).contiguous()
        sw = (sw * shrink).contiguous() if shrink != 1.0 else sw      # the seed part of the bound shrinks alike
        st.shard_ppr_begin(mn, mx, zmax, mass, 0.05, (sv, sw, scnt), flags, 0.5, 20, lay.n_groups, bufs)
        for i in range(20):
            for g in range(lay.n_groups):
                st.shard_ppr_sweep(i, g)
        st.shard_finish(mn, mx, flags, 50)
        torch.cuda.synchronize()
        got[label] = flags.cpu().numpy()
    eng.close()
    assert np.all(got["correct"] == 0)
    assert np.all(got["violated"] & FLAG_FP8_SATURATED), got["violated"]


def test_the_device_extends_exactly_when_the_measured_residual_is_above_the_tolerance(gpu_device):
    """The decision is taken from the residual the stage's final sweep MEASURED (csrc/ppr8.hip ppr8_decide_kernel), not
    from a prediction: a tolerance just above what 20 sweeps leave costs no extra sweep, one just below it buys
    extension stages until the measured residual is under it."""
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    b = 130
    kg, pb, fb, qf, qp = _small_engine_inputs(b, gpu_device)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pb, fb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=b, max_topk=50) as eng:
        idx, sc = eng.score_facts(qf, k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        # the base: 20 sweeps under a tolerance nothing exceeds (round 5: a call WITH a tolerance runs the stage plan whose
        # final measure reads lowest, a fixed count the one with a boundary fewer -- csrc/shard.hip ppr8_plan -- so the two
        # kinds of call agree to the parity bar, not bit for bit)
        fixed = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50, ppr_tol=1.0, ppr_max_iters=30)
        count = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50)
        torch.cuda.synchronize()
        assert eng.timings()["slab_width"] == 128
        r20 = float(fixed.residual.max())
        assert r20 > 0 and int(fixed.iters_used.max()) == 20 and int(count.iters_used.max()) == 20
        same = count.doc_idx == fixed.doc_idx
        assert float(same.float().mean()) > 0.99
        assert float(((count.doc_score - fixed.doc_score).abs() / fixed.doc_score)[same].max()) < 5e-6
        above = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50, ppr_tol=1.05 * r20, ppr_max_iters=30)
        below = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50, ppr_tol=0.95 * r20, ppr_max_iters=30)
        torch.cuda.synchronize()
        assert int(above.iters_used.max()) == 20 and np.all(above.flags.cpu().numpy() == 0)
        assert torch.equal(above.doc_score, fixed.doc_score) and torch.equal(above.doc_idx, fixed.doc_idx)
        assert torch.equal(above.residual, fixed.residual)
        used = int(below.iters_used.min())
        assert used == int(below.iters_used.max()) and used in (21, 23, 26, 29)       # whole stages of 1, 2, 3, 3 sweeps
        assert float(below.residual.max()) <= 0.95 * r20 and np.all(below.flags.cpu().numpy() == 0)


@pytest.mark.parametrize("b", [1, 8, 40])
def test_the_fp16_states_extend_on_the_device_then_flag_and_leave_the_repeat_to_the_caller(gpu_device, b):
    """The contract on the two-stage fp16 states (B <= 8: ppr_sv.hip; 9 .. 64: ppr16.hip), round 4: like the fp8 state they
    extend ON THE DEVICE -- stages of 1, 2, 3, 3 plain correction sweeps, each closed by a passage-row final sweep that
    measures again, gated by the control words a 1-block decision kernel sets (csrc/engine.hip ppr16_run) -- up to 9 sweeps
    beyond ppr_iters; a query still above the tolerance then gets HRAG_FLAG_NOT_CONVERGED (never silently unconverged
    scores for a raw caller of the C ABI) and HippoRAGEngine.retrieve_converged repeats it with the sweeps its residual
    asks for.  Ring graph: 20 sweeps leave ~1e-3, 29 leave ~2e-6 (some queries above the tolerance, some below): the former are
    flagged and repeated; everybody ends inside the bar with margin."""
    import torch
    from hipporag_amd._lib import FLAG_NOT_CONVERGED
    from hipporag_amd.engine import HippoRAGEngine
    n, src, dst, w, pv, pinned = _ring()
    csr, pass_bits, fact_bits, index = _index(n, src, dst, w, pv, 64, seed=11)
    n_p, tol = len(pv), 1.5e-6
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        fixed = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p)
        raw = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p, ppr_tol=tol, ppr_max_iters=400)
        short = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p, ppr_tol=tol, ppr_max_iters=23)
        torch.cuda.synchronize()
        assert eng.timings()["slab_width"] != 128         # NOT the fp8 state
        assert np.all(fixed.iters_used.cpu().numpy() == 20) and np.all(fixed.flags.cpu().numpy() == 0)   # tolerance 0
        raw_flags, raw_res = raw.flags.cpu().numpy(), raw.residual.cpu().numpy()
        assert np.all(raw.iters_used.cpu().numpy() == 29)                      # every extension stage ran ...
        assert np.all(((raw_flags & FLAG_NOT_CONVERGED) != 0) == (raw_res > tol))   # ... and whoever is still above the
        assert b == 1 or (raw_flags & FLAG_NOT_CONVERGED).any()                # tolerance says so (the ring after 29 sweeps:
                                                                              # 4e-7 .. 1.5e-4 from query to query)
        write_test_report(f"fp16_state_extension_ring_b{b}", {"residual_max_after_20": float(fixed.residual.max()),
                                                              "residual_max_after_29": float(raw_res.max()),
                                                              "residual_min_after_29": float(raw_res.min()),
                                                              "flagged": int(((raw_flags & FLAG_NOT_CONVERGED) != 0).sum())})
        assert float(raw_res.max()) < 0.1 * float(fixed.residual.max())        # nine more sweeps (the worst passage changes
                                                                              # from sweep to sweep on the ring: 0.03 .. 0.06)
        assert np.all(short.iters_used.cpu().numpy() == 23)                    # ppr_max_iters bounds the extension
        out = eng.retrieve_converged(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p, ppr_tol=tol,
                                     ppr_max_iters=400)
        torch.cuda.synchronize()
    assert np.all(out.flags.cpu().numpy() == 0) and float(out.residual.max()) <= tol
    got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
    worst = 0.0
    for q in range(0, b, max(1, b // 6)):
        want = oracle.retrieve_one(index, qf[q], qp[q]).x[index.passage_vertex]
        allow = prior_noise_allowance(index, qp[q])       # q = 12: the oracle's OWN two dot variants differ by 1.5e-5
        worst = max(worst, float((np.abs(got_sc[q] / want[got_idx[q]] - 1) - allow[got_idx[q]]).max()))
    assert worst < 1e-5 / 1.5, (b, worst)


@pytest.mark.parametrize("b", [4, 48])
def test_the_fp16_states_stop_extending_as_soon_as_the_measured_residual_is_under_the_tolerance(gpu_device, b):
    """The device decision on the fp16 states, like the fp8 one: a tolerance just above what the base sweeps leave costs
    nothing (bit-identical to the fixed count), one just below it buys whole extension stages until the MEASURED residual
    is under it, and nothing is flagged."""
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    # the ring: 20 sweeps leave a residual ~1e-3, far above the ~1e-7 the fp16 rounding of the correction puts under it
    n, src, dst, w, pv, pinned = _ring()
    csr, pass_bits, fact_bits, index = _index(n, src, dst, w, pv, 64, seed=11)
    qf = _bf16(synth.make_queries_np(fact_bits, b, seed=5)[0], gpu_device)
    qp = _bf16(synth.make_queries_np(pass_bits, b, seed=6)[0], gpu_device)
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=50) as eng:
        idx, sc = eng.score_facts(qf, k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        fixed = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50)
        torch.cuda.synchronize()
        assert eng.timings()["slab_width"] != 128
        r20 = float(fixed.residual.max())
        assert r20 > 1e-5 and int(fixed.iters_used.max()) == 20
        above = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50, ppr_tol=1.05 * r20, ppr_max_iters=30)
        below = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50, ppr_tol=0.2 * r20, ppr_max_iters=30)
        torch.cuda.synchronize()
        assert int(above.iters_used.max()) == 20 and np.all(above.flags.cpu().numpy() == 0)
        assert torch.equal(above.doc_score, fixed.doc_score) and torch.equal(above.doc_idx, fixed.doc_idx)
        used = int(below.iters_used.min())
        assert used == int(below.iters_used.max()) and used in (21, 23, 26), used       # whole stages of 1, 2, 3 sweeps
        assert float(below.residual.max()) <= 0.2 * r20 and np.all(below.flags.cpu().numpy() == 0)


def test_on_a_bipartite_graph_the_passage_only_measure_still_reads_every_sweep(gpu_device):
    """The convergence measure sees the PASSAGE rows only (include/hrag.h, hrag_retrieve).  Round 3 documented a blind
    spot for bipartite graphs -- passages on one side, entities on the other, NO entity-entity edge: mass moves between
    the sides, so "the passage rows change on alternate sweeps only and a single sweep's update can read 0".  That is
    true of the SERIES terms but not of the iteration the engine runs: it starts at x_0 = v, so x_k carries the
    trailing term (aP)^k v, and the passage rows move on EVERY sweep -- by (aP)^k v after an even sweep and by
    -a (aP)^(k-1) v after an odd one (CPU emulation: the measure reads 2.9 .. 3.2x the true error at sweeps 19, 20, 21,
    22 alike: the (1 + a) / (1 - a) = 3 of a purely oscillating mode, i.e. pessimistic = safe).  This test pins it on
    the device for an even and an odd count: the residual stays of the size of the true error (whose floor here is the
    ~5e-7 of stage rounding no convergence measure sees), and it contracts by ~a from 20 to 21 instead of collapsing."""
    import dataclasses
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    rng = np.random.default_rng(5)
    n_p, n_e = 256, 768
    n = n_p + n_e
    pv = np.arange(n_e, n)                                   # passages last, like the reference's vertex order
    src = np.repeat(pv, 6)
    dst = rng.integers(0, n_e, len(src))                     # passage -- entity edges only
    csr, pass_bits, fact_bits, index = _index(n, src, dst, np.ones(len(src)), pv, 64, seed=3)
    b = 70
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    res, err = {}, {}
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        for iters in (20, 21):
            out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=iters, k=n_p)
            torch.cuda.synchronize()
            assert eng.timings()["slab_width"] == 128 and np.all(out.flags.cpu().numpy() == 0)
            got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
            resid = out.residual.cpu().numpy()
            worst = ratio = 0.0
            for q in range(0, b, 10):
                want = oracle.retrieve_one(index, qf[q], qp[q]).x[index.passage_vertex]
                allow = prior_noise_allowance(index, qp[q])
                e = float((np.abs(got_sc[q] / want[got_idx[q]] - 1) - allow[got_idx[q]]).max())
                worst = max(worst, e)
                # no serious under-reading: the true error holds ~5e-7 of fp8-stage rounding the measure cannot see
                assert resid[q] > 0.5 * (e - 5e-7), (iters, q, resid[q], e)
            res[iters], err[iters] = float(resid.max()), worst
    write_test_report("bipartite_measure", {"residual_by_sweeps": res, "true_rel_err_by_sweeps": err})
    assert 0.3 * res[20] < res[21] < 0.8 * res[20], (res, err)        # contracts by ~damping; no collapse on the odd count


@pytest.mark.parametrize("b", [65, 256])
@pytest.mark.parametrize("name", sorted(CASES))
def test_accelerated_stages_on_adversarial_graphs_under_the_contract(gpu_device, name, b):
    """HRAG_OPT_ACCEL on the graphs chosen to break the fp8 state, through the contract end to end (what the mirror
    runs when the flag is on): no value may leave the e4m3 range -- the stage scales rest on the max-norm bound of the
    Chebyshev polynomial, not on its spectral contraction -- or, if one ever does, retrieve_converged must fall back to
    the plain plan; every passage score ends inside the bar with the margin of the plain suite.  The ring and the star
    forest put eigenvalues AT +-damping, the ends of the interval the polynomial is built for."""
    import dataclasses
    import torch
    from hipporag_amd._lib import FLAG_FP8_SATURATED, FLAG_NOT_CONVERGED, OPT_ACCEL
    from hipporag_amd.engine import HippoRAGEngine
    make, damping, iters = CASES[name]
    n, src, dst, w, pv, pinned = make()
    csr, pass_bits, fact_bits, index = _index(n, src, dst, w, pv, 64, seed=11, pinned_facts=pinned)
    index = dataclasses.replace(index, damping=damping)
    n_p = len(pv)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    for i in range(len(pinned)):
        qf_bits[i] = fact_bits[i]
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    tol = 1.5e-6
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=n_p, flags=OPT_ACCEL) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        raw = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, damping=damping, ppr_iters=iters, k=n_p,
                           ppr_tol=tol, ppr_max_iters=400)
        torch.cuda.synchronize()
        assert eng.timings()["slab_width"] == 128
        raw_flags, raw_used = raw.flags.cpu().numpy(), raw.iters_used.cpu().numpy()
        out = eng.retrieve_converged(_bf16(qp_bits, gpu_device), idx, sc, cnt, damping=damping, ppr_iters=iters,
                                     k=n_p, ppr_tol=tol, ppr_max_iters=400)
        torch.cuda.synchronize()
        assert eng.opt_flags & OPT_ACCEL                   # a fall-back restores the flag
        got_idx, got_sc, flags = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy(), out.flags.cpu().numpy()
        resid = out.residual.cpu().numpy()
    assert np.all(raw_flags & ~(FLAG_NOT_CONVERGED | FLAG_FP8_SATURATED) == 0), np.unique(raw_flags)
    assert not (raw_flags & FLAG_FP8_SATURATED).any(), "the max-norm scale bound was violated"
    assert raw_used.max() <= 30
    assert np.all(flags == 0), np.unique(flags)
    assert np.all(resid <= tol) and np.all(resid >= 0)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    check = list(range(len(pinned))) + list(range(len(pinned), b, max(1, b // 12)))
    worst = 0.0
    for q in check:
        want = oracle.retrieve_one(index, qf[q], qp[q]).x[index.passage_vertex]
        full = np.empty(n_p)
        full[got_idx[q]] = got_sc[q]
        nz = want > 0
        allow = prior_noise_allowance(index, qp[q])
        worst = max(worst, float((np.abs(full[nz] / want[nz] - 1) - allow[nz]).max()))
        assert np.all(full[~nz] == 0), q
    assert worst < 1e-5 / 1.5, (name, b, worst)


@pytest.mark.parametrize("b", [1, 8, 40])
@pytest.mark.parametrize("name", ["ring", "stars", "barbell_wild_weights", "tiny_component", "sparse_power_law"])
def test_a_base_count_of_12_sweeps_under_the_contract_keeps_the_error_bound_on_the_fp16_states(gpu_device, name, b):
    """Round 6, the latency mode of the narrow batches (RetrievalConfig.ppr_base_iters_narrow; csrc/engine.hip split16):
    under a tolerance only 12 sweeps ALWAYS run on the two-stage fp16 states (K1 = 9 on h, the residual sweep, one
    correction sweep, the measuring final sweep) and the measured residual adds stages where the graph mixes slowly.
    The bound include/hrag.h states must hold query by query exactly as it does for the worst-case base count of 20:
    true relative error of EVERY passage <= max(HRAG_PPR_ERR_K * residual, floor of the fp16 state), against the exact
    fp64 solution (the reference's PRPACK iterates to 1e-10: HippoRAG.py:1736-1743)."""
    import dataclasses
    import torch
    from hipporag_amd._lib import PPR_ERR_FLOOR_F16, PPR_ERR_K
    from hipporag_amd.engine import HippoRAGEngine
    make, damping, _ = CASES[name]
    n, src, dst, w, pv, pinned = make()
    csr, pass_bits, fact_bits, index = _index(n, src, dst, w, pv, 64, seed=11, pinned_facts=pinned)
    index = dataclasses.replace(index, damping=damping)
    n_p = len(pv)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    for i in range(min(b, len(pinned))):
        qf_bits[i] = fact_bits[i]
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    tol = 1.5e-6
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        raw = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, damping=damping, ppr_iters=12, k=n_p,
                           ppr_tol=tol, ppr_max_iters=400)
        torch.cuda.synchronize()
        assert eng.timings()["slab_width"] == (64 if b > 8 else (1 if b == 1 else 8))      # an fp16 state served the call
        raw_used = raw.iters_used.cpu().numpy()
        out = eng.retrieve_converged(_bf16(qp_bits, gpu_device), idx, sc, cnt, damping=damping, ppr_iters=12,
                                     k=n_p, ppr_tol=tol, ppr_max_iters=400)
        torch.cuda.synchronize()
        got_idx, got_sc, flags = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy(), out.flags.cpu().numpy()
        resid, used = out.residual.cpu().numpy(), out.iters_used.cpu().numpy()
    assert raw_used.min() >= 12 and raw_used.max() <= 12 + 9             # 12 always ran; at most the four extension stages
    assert np.all(flags == 0) and np.all(resid <= tol) and np.all(resid >= 0)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    worst = 0.0
    for q in sorted(set(list(range(min(b, len(pinned)))) + list(range(0, b, max(1, b // 6))))):
        exact = oracle.retrieve_one(index, qf[q], qp[q])
        want = exact.x[index.passage_vertex]
        full = np.empty(n_p)
        full[got_idx[q]] = got_sc[q]
        nz = want > 0
        allow = prior_noise_allowance(index, qp[q])
        e = float((np.abs(full[nz] / want[nz] - 1) - allow[nz]).max())
        worst = max(worst, e)
        assert e <= max(PPR_ERR_K * float(resid[q]), PPR_ERR_FLOOR_F16), (name, b, q, e, float(resid[q]))
    write_test_report(f"base12_contract_{name}_b{b}", {"worst_rel_err": worst, "residual_max": float(resid.max()),
                                                       "sweeps_first_call_max": int(raw_used.max()), "sweeps_max": int(used.max())})
    assert worst < 1e-5 / 1.5, (name, b, worst)
