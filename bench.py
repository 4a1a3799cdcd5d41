#!/usr/bin/env python
"""bench.py -- HippoRAG retrieval hot path on MI355X: queries/s + PPR SpMM roofline.

One "step" = one batch of B queries through the whole hot path with inputs resident in HBM:
  phase A (fact GEMM + min/max + top-5) -> identity filter -> phase B (passage GEMM + min/max,
  seeds, teleport, 20 PPR sweeps, normalise + gather + top-200).
Workload (BASELINE.json: the metric is quoted on the 1M-node KG): configs[2] =
  synthetic 1M-node / 10M-edge KG, 1M x 768 bf16 embeddings, batch 256, 20 PPR iterations.

    python bench.py [--gpus N --steps K --warmup W] [--config cfg2|cfg3|cfg4]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, external launcher)

N > 1 without a launcher (no WORLD_SIZE in the environment): bench.py spawns its own N ranks, one per GPU
(hipporag_amd/launch.py), and prints rank 0's line: `value` = the row-sharded leg (BASELINE.json's layout) when it is
parity-green, `value_hybrid` / `value_replica` beside it, and -- on the default configs[2] workload -- `configs3_strong`
= BASELINE configs[3] (global batch 1024 on the same index).  Prints ONE JSON line (rank 0), last.
"""

from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (V, E, D, B, seed)   -- SURVEY.md section 8 table; seed = 1234 + cfg number
    # configs[0] is the MuSiQue 1k-passage corpus (no dataset here, no network): SURVEY.md 8(d)'s synthetic
    # substitute of the same shape -- 1 000 passages, ~8k vertices -- labelled as such
    "cfg1s": dict(V=8_000, E=80_000, D=768, B=64, seed=1235,
                  label="configs[0] SUBSTITUTE: synthetic 1 000-passage / 8k-node / 80k-edge KG, 8k x 768 bf16, batch 64"),
    "cfg2": dict(V=100_000, E=1_000_000, D=768, B=64, seed=1236,
                 label="configs[1]: synthetic 100k-node/1M-edge KG, 100k x 768 bf16, batch 64"),
    "cfg3": dict(V=1_000_000, E=10_000_000, D=768, B=256, seed=1237,
                 label="configs[2]: synthetic 1M-node/10M-edge KG, 1M x 768 bf16, batch 256"),
    # NOT a BASELINE configuration: configs[2]'s sizes on a graph WITH locality (synth.make_kg(community=512): what
    # indexing a corpus document by document produces) -- shows what the sweep does when the gathers can hit the L2
    "cfg3loc": dict(V=1_000_000, E=10_000_000, D=768, B=256, seed=1237, community=512,
                    label="NON-BASELINE variant of configs[2]: 1M-node/10M-edge KG with community structure (512-entity communities, 90 % local edges), 1M x 768 bf16, batch 256"),
    # ... and the same graph under the REFERENCE's vertex numbering (entity ids in hash order): what the engine's
    # locality numbering (--locality auto: graph.locality_order) has to recover
    "cfg3hash": dict(V=1_000_000, E=10_000_000, D=768, B=256, seed=1237, community=512, hash_order=True,
                     label="NON-BASELINE variant of configs[2]: the community-structured 1M-node/10M-edge KG with its entity ids shuffled (the reference's hash-order numbering), 1M x 768 bf16, batch 256"),
    # one GPU's share of configs[4] (10M-node power-law KG, 10M x 1024 fp16 embeddings, 4096 / 8 queries)
    "cfg5gpu": dict(V=10_000_000, E=100_000_000, D=1024, B=512, seed=1239, power_law=True, fp16=True,
                    label="configs[4] per-GPU share: synthetic 10M-node/100M-edge power-law KG, 10M x 1024 fp16, batch 512"),
    # configs[3] itself: the 1M-node KG sharded across the GPUs of ONE node, GLOBAL batch 1024 whatever N is (strong
    # scaling: 128 queries per GPU at N = 8).  N = 1: the whole batch on one GPU.  N > 1 (`--gpus N`): `value` = the
    # row-sharded leg when parity-green (else hybrid), `value_hybrid` / `value_replica` beside it
    "cfg4": dict(V=1_000_000, E=10_000_000, D=768, B=1024, seed=1237, global_batch=1024,
                 label="configs[3]: synthetic 1M-node/10M-edge KG sharded across the GPUs of one node, global batch 1024"),
    # one GPU's share of configs[3]: shard 0 of the 8-way row shard of the 1M-node KG with the GLOBAL batch of
    # 1024 queries (compute of one GPU; the exchanges need the other 7 GPUs and are not performed)
    "cfg4gpu": dict(V=1_000_000, E=10_000_000, D=768, B=1024, seed=1237, shard_of=8,
                    label="configs[3] per-GPU share: shard 0 of 8 row shards of the 1M-node/10M-edge KG, global batch 1024"),
    # configs[3] with its RESULT checked: all 8 row shards as threads on ONE device (dist.LocalComm: the emulated
    # gather of SURVEY.md 8e), the global batch of 1024, compared with the CPU oracle and the single-GPU engine
    "cfg4local": dict(V=1_000_000, E=10_000_000, D=768, B=1024, seed=1237, shard_of=8, local_shards=True,
                      label="configs[3] parity run: all 8 row shards of the 1M-node/10M-edge KG emulated on ONE device, global batch 1024"),
    "tiny": dict(V=20_000, E=200_000, D=256, B=32, seed=1235, label="tiny smoke workload"),
    # NOT a BASELINE configuration: a graph with REAL topology (tools/real2wiki.py: a deterministic triple extractor over
    # the 6 119-passage 2WikiMultihopQA corpus the reference ships, tools/make_real2wiki.py), 32 disjoint copies with
    # interleaved ids = 1.5M vertices / 13.1M entries / 4.2M facts, the reference's mock embedding recipe (64-d uniform).
    # `--locality auto` vs none shows what the graph compiler's numbering buys on real per-document locality
    "real2wiki": dict(real2wiki=True, tiles=32, V=0, E=0, D=64, B=256, seed=1240,
                      label="NON-BASELINE: real-topology KG from the 2WikiMultihopQA corpus (LLM-free extractor), 32 "
                            "interleaved disjoint copies: 1.5M vertices / 6.55M edges / 4.2M facts, 64-d mock embeddings, batch 256"),
    "real2wiki1": dict(real2wiki=True, tiles=1, V=0, E=0, D=64, B=256, seed=1240,
                       label="NON-BASELINE: real-topology KG from the 2WikiMultihopQA corpus (LLM-free extractor) at its own "
                             "size: 46.9k vertices / 205k edges / 130k facts, 64-d mock embeddings, batch 256"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
PPR_ITERS, K_F, K_P, DAMPING, PASSAGE_W = 20, 5, 200, 0.5, 0.05


def spmm_algorithmic_bytes(nnz, V, Np, B, state_bytes=4):
    """SURVEY.md 8(d), per sweep of a batch of B vectors: CSR once + read x + write y + read the
    passage-dense teleport term, `index 4 B, value 4 B, state 4 B`.  That figure (state_bytes=4) is
    what one PPR iteration of the reference algorithm has to move and is what `roofline.achieved`
    prices; state_bytes=2 / 1 give the same formula at the width the fp16 / fp8 kernels store."""
    return nnz * 8 + (V + 1) * 4 + 2 * V * B * state_bytes + Np * B * 4


def sim_algorithmic_bytes(F, Np, D, B):
    return (F + Np) * D * 2 + 2 * B * D * 2 + B * Np * 4


def load_traffic():
    """HBM bytes per SpMM launch from the PMC passes, if tools/pmc_summary.py has produced them."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


def _time_launches(fn, n_l):
    import torch
    fn(4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(n_l)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_l


def fp8_mode_counts(iters, damping=None):
    """Launches of one retrieve by kernel instantiation "<mode>" or "<mode>/<residual form>": ppr8_plan and the
    residual-form schedule of ppr8_begin in csrc/shard.hip (engine.fp8_stage_plan: 20 = 1+2+3+4+4+4+2; the residual
    travels in its 3-byte form once damping^k <= 2^-6)."""
    from hipporag_amd.engine import fp8_stage_plan
    damping = DAMPING if damping is None else damping
    stages = fp8_stage_plan(iters, damping)
    counts = {"C": sum(m - 1 for m in stages[1:]), "B0": 1}
    k, r16 = 0, False
    for si, m in enumerate(stages):
        k += m
        if si == 0:
            continue
        if si + 1 < len(stages):
            out16 = damping ** k <= 1.0 / 64.0
            key = f"B/{(1 if r16 else 0) | (2 if out16 else 0)}"
            counts[key] = counts.get(key, 0) + 1
            r16 = out16
        else:
            counts[f"F/{1 if r16 else 0}"] = 1
    return counts


def measure_roofline(eng, kg, V, B, phases, config_name, n_l):
    """Dominant kernel (the PPR sweep): average launch duration (HIP events on the launch stream, n_l
    back-to-back launches per kernel instantiation) against SURVEY.md 8(d)'s algorithmic bytes per PPR
    iteration.  On the fp8 path one retrieve launches ppr8_kernel in four instantiations (C stage sweep,
    B boundary, B0 first boundary, F final): `achieved` / `frac` price the AVERAGE launch of a retrieve
    (counts from the stage plan), the C-only figure is kept beside it.  Returns (roofline dict, f8, f16)."""
    f8 = phases["slab_width"] == 128                # hrag_retrieve took the staged fp8-state path
    f16 = phases["slab_width"] == 64 and B > 32     # ... the two-stage fp16-state path
    per_mode = None
    harness_ms = None
    replay_ms = replay_err = None
    if f8:
        counts = fp8_mode_counts(PPR_ITERS)
        per_mode = {}
        for key in counts:
            mode, _, rio = key.partition("/")
            per_mode[key] = _time_launches(lambda n, mode=mode, rio=int(rio or 0): eng.ppr_sweeps(
                B, n, DAMPING, f8=True, f8_mode=mode, f8_rio=rio), n_l)
        harness_ms = sum(counts[m] * per_mode[m] for m in counts) / PPR_ITERS
        main_ms = _time_launches(lambda n: eng.ppr_sweeps(B, n, DAMPING, main_only=True, f8=True), n_l)
        # the ceiling of the formulation, measured in THIS run on the engine's own matrix and state: the state-row
        # gathers of a stage sweep alone (256 bytes per matrix slot, padding slots included; ppr8_pair_replay_kernel)
        try:
            replay_ms = _time_launches(lambda n: eng.ppr_sweeps(B, n, DAMPING, f8=True, f8_gather_replay=True), n_l)
        except Exception as exc:
            replay_ms, replay_err = None, f"{type(exc).__name__}: {exc}"
        # The average launch of a retrieve, measured IN SITU: HIP events (inside the library, on the launch stream) around
        # the PPR_ITERS sweep launches of real retrieves on fresh queries (median of the profiled steps).  The per-
        # instantiation figures above come from a harness that re-launches ONE instantiation on whatever the state holds;
        # they are a breakdown, not the measurement: on hub-heavy graphs the harness's boundary launches drift into the
        # e4m3 saturation path (atomics) and read up to 2x slow (real2wiki, round 5), which the in-situ figure cannot
        spmm_ms = phases["ppr_ms"] / PPR_ITERS
    else:
        main_ms = _time_launches(lambda n: eng.ppr_sweeps(B, n, DAMPING, main_only=True, f16=f16), n_l)
        spmm_ms = _time_launches(lambda n: eng.ppr_sweeps(B, n, DAMPING, main_only=False, f16=f16), n_l)
    nnz = kg.csr.nnz
    sb = 1 if f8 else 2 if f16 else 4
    # one launch = one sweep (PPR iteration) of the whole batch: SURVEY.md 8(d)'s per-iteration bytes
    alg = spmm_algorithmic_bytes(nnz, V, kg.n_passages, B, 4)
    alg_stored = spmm_algorithmic_bytes(nnz, V, kg.n_passages, B, sb)
    achieved = alg / (spmm_ms * 1e-3) / 1e9
    traffic = load_traffic()
    bc, n_slabs = (128, (B + 127) // 128) if f8 else (64, (B + 63) // 64) if f16 else eng.layout(B)
    kernel = "ppr8_kernel" if f8 else "ppr16_kernel" if f16 else "ppr_spmm_kernel"   # key of pmc_traffic.json
    # slab pairs are swept by ppr8_pair_kernel (two slabs per wavefront), an odd last slab by ppr8_kernel
    kernel_launched = ("ppr8_pair_kernel" + (" + ppr8_kernel (odd last slab)" if n_slabs % 2 else "")
                       if f8 and n_slabs >= 2 else kernel)
    ppr_iter_ms = phases["ppr_ms"] / PPR_ITERS      # every kernel of the PPR stage (init, reduce) / iterations
    tr = (traffic or {}).get(kernel, {})
    traffic_bytes = tr.get("bytes_per_launch") if tr.get("workload") == f"{config_name}:B{B}" else None
    # the PMC figures are replayed, not measured in this run: they stand only while the kernel's source is the one they
    # were collected on (tools/prof_summary.py stamps its hash) -- a changed kernel reports traffic = null, not a stale number
    traffic_note = None
    if traffic_bytes is not None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from prof_summary import kernel_source_sha16
        now = kernel_source_sha16(kernel)
        if tr.get("kernel_source_sha16") != now:
            traffic_bytes = None
            traffic_note = (f"profiles/pmc_traffic.json was collected on another build of the kernel source "
                            f"({tr.get('kernel_source_sha16')} != {now}): re-run tools/gpu_profile.sh")
    if f8 and traffic_bytes is not None and tr.get("by_instantiation"):
        # average launch of one retrieve, like `achieved`: instantiation <mode, residual form> weighted by the plan
        num = {"C": 0, "B": 1, "F": 2, "B0": 3}
        tot = cnt = 0
        for key, c in counts.items():
            mode, _, rio = key.partition("/")
            base_key = f"<{num[mode]},{int(rio or 0)}"
            # the final sweep of a retrieve that reports its residual is the est-measuring instantiation
            inst = (tr["by_instantiation"].get(base_key + ",est>") if mode == "F" else None) or tr["by_instantiation"].get(base_key + ">")
            if inst:
                tot += c * inst["bytes_per_launch"]
                cnt += c
        if cnt == PPR_ITERS:
            traffic_bytes = tot / cnt
    roofline = {
        "bound": "hbm", "kernel": kernel_launched, "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "frac_definition": ("algorithmic bytes of one PPR iteration / average sweep launch of a real retrieve: HIP events around "
                            "its 20 sweep launches (all instantiations of the stage plan), median of the profiled steps"
                            if f8 else "average launch of the sweep kernel (+ its long-row reduce)"),
        "launch_ms_from_instantiation_harness": harness_ms,
        "instantiation_harness_agrees": (bool(abs(harness_ms / spmm_ms - 1) < 0.05) if harness_ms else None),
        # PMC traffic is only meaningful for the workload it was collected on (profiles/pmc_traffic.json)
        "traffic": traffic_bytes,
        "traffic_source": traffic_note or ("replayed from profiles/pmc_traffic.json (separate rocprofv3 --pmc passes of this command, "
                                           "FETCH_SIZE x2-corrected + WRITE_SIZE, stamped with the hash of the kernel source they were "
                                           "collected on and dropped when it changes); not measured in this run"),
        "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_definition": "SURVEY 8(d): nnz*8 + (V+1)*4 + 2*V*B*4 + Np*B*4 per PPR iteration",
        "state_bytes_stored": sb, "algorithmic_bytes_at_stored_state_width": alg_stored,
        "frac_at_stored_state_width": alg_stored / (spmm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "gather_bytes_per_launch": nnz * B * sb,
        "launch_ms": spmm_ms, "launch_ms_main_kernel_only": main_ms,
        "launch_ms_by_mode": per_mode, "launches_by_mode": fp8_mode_counts(PPR_ITERS) if f8 else None,
        "frac_mode_c": (alg / (per_mode["C"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if f8 else None,
        "ppr_stage_ms_per_iteration": ppr_iter_ms,
        "frac_whole_ppr_stage": alg / (ppr_iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        # "0.37 of 0.42": a sweep that fetches one state row per matrix slot cannot run faster than the replay of exactly
        # those fetches; frac_ceiling = what `frac` would read if every sweep launch took only that long
        "gather_replay_ms": replay_ms,
        "gather_replay_definition": ("hrag_ppr_sweeps flag 256, same run: the gathers of a stage sweep alone -- the engine's SELL-8 "
                                     "(col, val) stream and one 256-byte piece of the e4m3 state per slot; nothing computed or stored"
                                     if f8 else None),
        "gather_replay_error": replay_err,
        "formulation_ceiling_frac": (alg / (replay_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if replay_ms else None,
        "frac_of_formulation_ceiling": (replay_ms / spmm_ms) if replay_ms else None,
        "stage_sweep_over_gather_replay": (per_mode["C"] / replay_ms) if (replay_ms and per_mode) else None,
        "launches_timed": n_l, "slab_width": bc, "n_slabs": n_slabs,
        "frac_of_measured_copy_peak_6290": achieved / 6290.0,
    }
    return roofline, f8, f16


def device_reset_vectors(eng, kg, q_fact, q_pass, cnt, n_q):
    """The reset ("personalization") vectors the DEVICE built for the first n_q queries of a batch, as fp64 host arrays
    [n_q][V] in the caller's vertex numbering: phase A again (deterministic), then the stage-level operators of
    include/hrag.h -- hrag_stage_seeds, hrag_sim_scores + hrag_row_minmax + hrag_stage_teleport -- which launch the SAME
    kernels the fused hrag_retrieve launches (build_seeds_kernel, rows_to_slab_kernel<kMinMaxScale>), so the values are the
    ones the PPR solve of that batch started from.  What the PPR-only parity figure is computed on (HippoRAG.py:1736-1749:
    the step the reference hands to PRPACK)."""
    import torch
    from hipporag_amd.engine import EngineStages
    st = EngineStages(eng)
    idx, sc = eng.score_facts(q_fact, k=K_F)
    sv, sw, scnt, flags = st.seeds(idx, sc, cnt, K_F)
    raw = st.sim_scores("passages", q_pass)
    mn, mx = st.row_minmax(raw)
    tele = st.teleport(raw, mn, mx, PASSAGE_W, flags)              # [slab][passage][column]
    torch.cuda.synchronize()
    bc = tele.shape[2]
    inv = eng._inv_perm.cpu().numpy() if getattr(eng, "_perm", None) is not None else None   # engine numbering -> caller's
    pv = np.asarray(kg.passage_vertex)                  # teleport rows are in passage order (unchanged by a renumbering)
    out = []
    sv_h, sw_h, sc_h = sv.cpu().numpy(), sw.cpu().numpy().astype(np.float64), scnt.cpu().numpy()
    for q in range(n_q):
        v = np.zeros(eng.num_vertices)
        v[pv] = tele[q // bc, :, q % bc].double().cpu().numpy()
        ids = sv_h[q, :sc_h[q]]
        ids = inv[ids] if inv is not None else ids
        np.add.at(v, ids, sw_h[q, :sc_h[q]])
        out.append(v)
    return out


def cpu_baseline(kg, fact_emb_t, pass_emb_t, qf_t, qp_t, gpu_idx, gpu_scores, budget_s, max_queries,
                 vec_queries=32, nx_budget_s=6.0, also=None, device_resets=None, ppr_only_budget_s=25.0):
    """Reference-style CPU loop on a bounded sample (rank 0 only) + parity spot check.
    also: {name: (ids, scores)} -- further device results for the same queries, checked against the same oracle rows.
    device_resets: device_reset_vectors() of the same batch -> the PPR-ONLY error of every leg (its scores against the
    exact fp64 solution for the reset vector the device itself built: no similarity / prior arithmetic in it)."""
    import oracle
    from oracle.checks import percentiles, ulp4_report
    from oracle.cpu_baseline import ReferenceStyleRetriever
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([d.get("num_threads", 1) for d in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    t_prep = time.perf_counter()
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    p = oracle.column_normalize(a)
    index = oracle.RefIndex(fact_emb=fact_emb_t.float().cpu().numpy(), passage_emb=pass_emb_t.float().cpu().numpy(),
                            subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex, num_chunks=kg.num_chunks,
                            passage_vertex=kg.passage_vertex, p=p)
    ref = ReferenceStyleRetriever(index)
    prep_s = time.perf_counter() - t_prep
    qf = qf_t.float().cpu().numpy()
    qp = qp_t.float().cpu().numpy()
    n_done, t0 = 0, time.perf_counter()
    ids_equal, max_rel = True, 0.0
    exact_pos = n_pos = exact_rows = 0
    from oracle.checks import ranked_parity
    tie_window = 0.0
    also_stats = {}
    rel_all, ulp4 = [], {"equal": True, "exact": 0, "n": 0}

    def rel_errs(ids_row, sc_row, full_scores):
        want = np.asarray(full_scores, dtype=np.float64)[ids_row]
        nz = want > 0
        return np.abs(np.asarray(sc_row, dtype=np.float64)[nz] / want[nz] - 1)

    while n_done < min(max_queries, qf.shape[0]):
        ids, scores = ref.retrieve_one(qf[n_done], qp[n_done])
        g_ids = gpu_idx[n_done]
        full = np.empty(len(ids)); full[ids] = scores
        # permutations are tolerated only inside a tie window that follows the MEASURED score error (2.2 x the worst
        # relative deviation of this query's scores, at least 2e-6, at most 2e-5: tests/helpers.ranked_parity); how
        # many ranks needed that is reported beside the verdict
        rep = ranked_parity(g_ids, gpu_scores[n_done], ids, scores, full)
        ids_equal = ids_equal and rep["equal"]
        exact_pos += rep["exact_positions"]; n_pos += rep["n"]; exact_rows += int(rep["exact_positions"] == rep["n"])
        max_rel = max(max_rel, rep["worst_rel_err"])
        tie_window = max(tie_window, rep["rel_gap"])
        rel_all.append(rel_errs(g_ids, gpu_scores[n_done], full))
        u4 = ulp4_report(g_ids, ids, scores)       # the same verdict at SURVEY 8(c)'s own window (4 ulp of fp32)
        ulp4["equal"] = ulp4["equal"] and u4["equal"]; ulp4["exact"] += u4["exact_positions"]; ulp4["n"] += u4["n"]
        for name, (a_idx, a_sc) in (also or {}).items():
            r2 = ranked_parity(a_idx[n_done], a_sc[n_done], ids, scores, full)
            st = also_stats.setdefault(name, {"topk_ids_equal": True, "exact": 0, "n": 0, "max_rel_score_err": 0.0, "tie_window_rel": 0.0,
                                              "rel": [], "ulp4_equal": True})
            st["rel"].append(rel_errs(a_idx[n_done], a_sc[n_done], full))
            st["ulp4_equal"] = st["ulp4_equal"] and ulp4_report(a_idx[n_done], ids, scores)["equal"]
            st["topk_ids_equal"] = st["topk_ids_equal"] and bool(r2["equal"])
            st["exact"] += r2["exact_positions"]; st["n"] += r2["n"]
            st["max_rel_score_err"] = max(st["max_rel_score_err"], r2["worst_rel_err"])
            st["tie_window_rel"] = max(st["tie_window_rel"], r2["rel_gap"])
        n_done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    base = {
        "value": n_done / el, "unit": "queries/s", "cores": int(blas_threads), "kind": "port",
        "sample": f"{n_done} queries of the same workload, reference-style per-query loop "
                  f"(fp32 np.dot on {blas_threads} BLAS threads, Python seed loops, single-thread "
                  f"PRPACK Gauss-Seidel port tol 1e-10); {el:.1f} s (+{prep_s:.1f} s index prep)",
        "sim_s_per_query": ref.sim_time / max(n_done, 1), "ppr_s_per_query": ref.ppr_time / max(n_done, 1),
        "host_cpus": os.cpu_count(), "cpu_model": _cpu_model(),
    }
    parity = {"queries_checked": n_done, "topk_ids_equal": bool(ids_equal),
              "topk_ids_equal_definition": "identical ranked ids; a permutation is accepted only inside a run of "
                                           "oracle scores closer than tie_window_rel = max(2e-6, 2.2 x the measured "
                                           "max relative score error) (tie class)",
              "tie_window_rel": tie_window,
              "exact_id_fraction": exact_pos / max(n_pos, 1), "queries_with_identical_id_lists": exact_rows,
              "max_rel_score_err": max_rel,
              "rel_score_err": percentiles(np.concatenate(rel_all)) if rel_all else None,
              "rel_score_err_definition": "end to end: |device score / oracle score - 1| over every returned (query, rank)",
              # the contract's own tie window (set-equality only where adjacent oracle scores differ by
              # <= 4 ulp of fp32 = 4.8e-7 relative): how much of the verdict above leans on the wider measured-error window
              "topk_ids_equal_at_4ulp_window": bool(ulp4["equal"]), "tie_window_4ulp_rel": 4 * 2.0 ** -23}
    for name, st in also_stats.items():
        parity[name] = {"topk_ids_equal": st["topk_ids_equal"], "exact_id_fraction": st["exact"] / max(st["n"], 1),
                        "max_rel_score_err": st["max_rel_score_err"], "tie_window_rel": st["tie_window_rel"],
                        "rel_score_err": percentiles(np.concatenate(st["rel"])) if st["rel"] else None,
                        "topk_ids_equal_at_4ulp_window": bool(st["ulp4_equal"])}
    # ---- PPR-ONLY error: every leg's scores against the exact fp64 PPR of the reset vector the DEVICE built (the step
    # the reference hands to igraph / PRPACK, HippoRAG.py:1736-1749).  No similarity, min-max or seed arithmetic in it, so
    # -- unlike the end-to-end figure, whose maximum is owned by the fp32 prior -- it moves with the PPR plan
    if device_resets:
        t_p = time.perf_counter()
        legs = {"headline": (gpu_idx, gpu_scores)}
        legs.update(also or {})
        errs = {name: [] for name in legs}
        n_p = 0
        for qi, v in enumerate(device_resets[:n_done]):
            x = oracle.ppr_exact(index.p, v, DAMPING)[kg.passage_vertex]
            for name, (l_idx, l_sc) in legs.items():
                errs[name].append(rel_errs(l_idx[qi], l_sc[qi], x))
            n_p += 1
            if time.perf_counter() - t_p > ppr_only_budget_s:
                break
        parity["ppr_only"] = {
            "definition": "|device score / x*_p - 1| over every returned (query, rank), x* = exact fp64 PPR (oracle.ppr_exact) of "
                          "the reset vector the device built for that query (read back through hrag_stage_seeds / "
                          "hrag_stage_teleport: the kernels the fused call launches)",
            "queries_checked": n_p, "seconds": time.perf_counter() - t_p,
            **{name: percentiles(np.concatenate(e)) if e else None for name, e in errs.items()}}
    # ---- "vectorised" leg: batched sgemm + argpartition + OpenMP SpMM over all host cores, the
    # same algorithm and sweep count as the GPU path -- the ratio against THIS number is the one free of the
    # reference's Python overhead
    try:
        from oracle.cpu_baseline import VectorisedRetriever
        vb = int(min(vec_queries, qf.shape[0], VectorisedRetriever.MAX_B))
        vr = VectorisedRetriever(index)
        vr.retrieve(qf[:2], qp[:2], iters=2)                       # touch the library / BLAS once
        vr.sim_time = vr.seed_time = vr.ppr_time = vr.rank_time = 0.0
        t1 = time.perf_counter()
        v_ids, v_sc = vr.retrieve(qf[:vb], qp[:vb], iters=PPR_ITERS, k=gpu_idx.shape[1])
        v_el = time.perf_counter() - t1
        v_rel = np.abs(v_sc[:vb] - gpu_scores[:vb]) / np.maximum(gpu_scores[:vb], 1e-300)
        base["vectorised"] = {
            "value": vb / v_el, "unit": "queries/s", "cores": int(os.cpu_count() or 1), "blas_threads": int(blas_threads),
            "sample": f"one batch of {vb} queries of the same workload: fp32 sgemm (all BLAS threads), argpartition, "
                      f"numpy seeds, {PPR_ITERS}-sweep fp32 power iteration as an OpenMP SpMM on all cores; {v_el:.2f} s",
            "sim_s": vr.sim_time, "seed_s": vr.seed_time, "ppr_s": vr.ppr_time, "rank_s": vr.rank_time,
            "ids_equal_to_gpu_fraction": float((v_ids[:vb] == gpu_idx[:vb]).mean()),
            "max_rel_score_diff_to_gpu_at_same_rank": float(v_rel.max()),
        }
    except Exception as exc:   # a missing gcc / OpenMP must not cost the line
        base["vectorised"] = {"error": f"{type(exc).__name__}: {exc}"}
    # ---- networkx.pagerank(tol=1e-10) leg for the PPR step (the small configurations only: building a
    # 10M-edge networkx graph takes minutes and ~10 GB)
    if kg.num_vertices <= 200_000 and nx_budget_s > 0:
        try:
            from oracle.cpu_baseline import networkx_pagerank_leg
            resets = []
            for q in range(min(3, n_done)):
                r = oracle.retrieve_one(index, qf[q], qp[q])
                if r.reset is not None:
                    resets.append(r.reset)
            if resets:
                base["networkx_pagerank"] = networkx_pagerank_leg(index, np.array(resets), nx_budget_s)
        except Exception as exc:
            base["networkx_pagerank"] = {"error": f"{type(exc).__name__}: {exc}"}
    else:
        base["networkx_pagerank"] = {"skipped": "graph too large for a networkx build inside the bench budget"}
    return base, parity


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


class _NoPeers:
    """Stand-in for the exchange steps when only ONE shard of `world` runs (per-GPU share measurement):
    reductions see this shard's contribution only, gathers return `world` copies of it (so that the
    candidate merge does the full-size work), state exchanges are skipped."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def all_reduce(self, t, op):
        return t

    def all_gather(self, t):
        return [t] * self.world

    def exchange(self, buf, lay, g):
        return None

    def wait(self, handle):
        pass


def local_shards_parity(cfg, batch, cpu_queries, exchange_groups, dev):
    """BASELINE configs[3] with its result checked on one GPU: the `shard_of` row-shard engines run as threads of this
    process on one device (the state exchange is a barrier on shared buffers), the merged ranking of the global batch
    is compared with the fp64 CPU oracle (`cpu_queries` queries) and with the single-GPU engine on the relabelled
    index (bit-identity of every query).  The wall time serialises 8 GPUs' work on one and is NOT a multi-GPU rate.
    Returns the result line as a dict (`bench.py --config cfg4local` prints it; tests/test_gpu_full_size.py asserts on it)."""
    import torch
    import oracle
    from hipporag_amd import dist as hd, synth
    from hipporag_amd.engine import HippoRAGEngine
    from oracle.checks import tie_aware_report
    world, V, E, D, B, seed = cfg["shard_of"], cfg["V"], cfg["E"], cfg["D"], batch or cfg["B"], cfg["seed"]
    kg = synth.make_kg(V, E, seed)
    pass_emb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, dev)
    fact_emb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, dev)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    qf = synth.make_queries_torch(fact_emb, B, seed + 100)[0]
    qp = synth.make_queries_torch(pass_emb, B, seed + 500)[0]
    kw = dict(link_top_k=K_F, damping=DAMPING, passage_node_weight=PASSAGE_W, ppr_iters=PPR_ITERS, k=K_P)
    tm = {}
    # the long-row cut fixes the summation order of hub rows: the same explicit value on the shards and on the
    # unsharded engine they are compared with (hrag_opts.sell_seg_len; auto would pick 256 vs 2048 here)
    seg = 256
    got = hd.run_local_shards(world, sidx, pass_emb, fact_emb, qf, qp, kw, exchange_groups, dev, K_P, timings=tm,
                              sell_seg_len=seg)
    f_idx, f_sc, d_idx, d_sc, flags = got
    # ---- the single-GPU engine on the same relabelled index: bit-identity of the whole batch
    with HippoRAGEngine(sidx.csr, sidx.passage_vertex, pass_emb, fact_emb, sidx.subj_vertex, sidx.obj_vertex,
                        sidx.num_chunks, max_batch=B, max_topk=K_P, sell_seg_len=seg) as eng:
        i1, s1 = eng.score_facts(qf, k=K_F)
        o1 = eng.retrieve(qp, i1, s1, torch.full((B,), K_F, dtype=torch.int32, device=dev), **kw)
        torch.cuda.synchronize()
        one = tuple(t.cpu().numpy() for t in (i1, s1, o1.doc_idx, o1.doc_score))
    bit = {"fact_ids": bool(np.array_equal(f_idx, one[0])), "fact_scores": bool(np.array_equal(f_sc, one[1])),
           "doc_ids": bool(np.array_equal(d_idx, one[2])), "doc_scores": bool(np.array_equal(d_sc, one[3])),
           "doc_ids_equal_fraction": float((d_idx == one[2]).mean()),
           "doc_scores_max_rel_diff": float((np.abs(d_sc - one[3]) / np.maximum(one[3], 1e-30)).max()),
           "sell_seg_len": seg}
    # ---- the fp64 oracle on the ORIGINAL index (passage positions do not change under the relabelling)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    index = oracle.RefIndex(fact_emb=fact_emb.float().cpu().numpy(), passage_emb=pass_emb.float().cpu().numpy(),
                            subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex, num_chunks=kg.num_chunks,
                            passage_vertex=kg.passage_vertex, p=oracle.column_normalize(a))
    qf_h, qp_h = qf.float().cpu().numpy(), qp.float().cpu().numpy()
    n_q = max(1, min(cpu_queries, B))
    qs = sorted(set(np.linspace(0, B - 1, n_q).astype(int).tolist()))
    ok, worst, exact, npos = True, 0.0, 0, 0
    for q in qs:
        ref = oracle.retrieve_one(index, qf_h[q], qp_h[q])
        rep = tie_aware_report(d_idx[q], ref.sorted_doc_ids[:K_P], ref.sorted_doc_scores[:K_P], rel_gap=2e-5)
        ok = ok and rep["equal"]
        exact += rep["exact_positions"]; npos += rep["n"]
        want = ref.x[kg.passage_vertex][d_idx[q]]
        worst = max(worst, float((np.abs(d_sc[q] - want) / want).max()))
    wall = tm.get("wall_s_all_shards_on_one_device", float("nan"))
    return {
        "metric": "retrieval_queries_per_sec", "value": B / wall, "unit": "queries/s", "n_gpus": 1, "steps": 1,
        "warmup": 0, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["label"], "V": V, "E": E, "nnz": kg.csr.nnz, "global_batch": B,
                   "shards": world, "exchange_groups": exchange_groups, "ppr_iters": PPR_ITERS,
                   "parallelism": f"{world} row shards emulated as threads on ONE device (barrier exchange): a PARITY run; "
                                  "value is the serialised wall time of 8 GPUs' work, not a multi-GPU rate"},
        "parity_vs_oracle": {"queries_checked": len(qs), "queries": qs, "topk_ids_equal": bool(ok),
                             "exact_id_fraction": exact / max(npos, 1), "max_rel_score_err": worst,
                             "flags_or": int(np.bitwise_or.reduce(flags))},
        "bit_identical_to_single_gpu_engine_on_relabelled_index": bit,
    }


def bench_local_shards(args, cfg, dev):
    print(json.dumps(local_shards_parity(cfg, args.batch, args.cpu_queries, args.exchange_groups, dev)))
    return 0


def bench_shard_share(args, cfg, dev):
    """One GPU's compute share of a row-sharded configuration: shard 0 of `shard_of` with the global batch.
    The other shards' rows never arrive, so the scores are meaningless; the kernels, their sizes and the
    memory traffic are exactly those of one GPU of the sharded job (no collective is timed)."""
    import torch
    from hipporag_amd import dist as hd, synth
    from hipporag_amd.engine import ShardStages
    world, V, E, D, B, seed = cfg["shard_of"], cfg["V"], cfg["E"], cfg["D"], args.batch or cfg["B"], cfg["seed"]
    kg = synth.make_kg(V, E, seed)
    pass_emb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, dev)
    fact_emb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, dev)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    eng = hd.build_shard_engine(sidx, pass_emb, fact_emb, 0, B, K_P)
    rs = hd.ShardedRetriever(ShardStages(eng), _NoPeers(0, world), groups=args.exchange_groups)
    n = args.steps + args.warmup
    qf = [synth.make_queries_torch(fact_emb, B, seed + 100 + i)[0] for i in range(n)]
    qp = [synth.make_queries_torch(pass_emb, B, seed + 500 + i)[0] for i in range(n)]
    cnt = torch.full((B,), K_F, dtype=torch.int32, device=dev)

    def step(i):
        idx, sc = rs.score_facts(qf[i], k=K_F)
        idx = idx.clamp(min=0)      # foreign candidates are copies of the local ones here: keep the ids valid
        return rs.retrieve(qp[i], idx, sc, cnt, link_top_k=K_F, damping=DAMPING, passage_node_weight=PASSAGE_W,
                           ppr_iters=PPR_ITERS, k=K_P, check_saturation=False)    # one shard of 8: the scores are meaningless anyway

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n):
        step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    lay = eng.shard_layout(B, args.exchange_groups)
    rps = sidx.rows_per_shard
    nnz_own = int(sidx.csr.row_ptr[rps] - sidx.csr.row_ptr[0])
    wire = (world - 1) / world * sidx.num_vertices * 128 * lay.n_slabs
    # the share of SURVEY 8(d)'s per-iteration bytes this GPU owns: its rows of the CSR, of y and of v; all of x
    alg_share = nnz_own * 8 + (rps + 1) * 4 + (sidx.num_vertices + rps) * B * 4 + (kg.n_passages // world) * B * 4
    ms = el * 1e3 / max(args.steps, 1)
    result = {
        "metric": "retrieval_queries_per_sec", "value": B * args.steps / el, "unit": "queries/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["label"], "V": V, "E": E, "nnz": kg.csr.nnz, "global_batch": B,
                   "shard": f"0 of {world}", "rows_per_shard": rps, "nnz_this_shard": nnz_own,
                   "exchange_groups": int(lay.n_groups), "ppr_iters": PPR_ITERS,
                   "parallelism": f"compute share of one of {world} row shards; exchanges NOT performed (1 GPU)"},
        "note": "value = global batch / compute time of ONE shard: an upper bound of the sharded job's rate "
                "(exchange time comes on top: wire_bytes_received_per_gpu_per_sweep over xGMI)",
        "wire_bytes_received_per_gpu_per_sweep": wire, "state_bytes_per_buffer": int(lay.state_bytes),
        "algorithmic_bytes_per_iteration_this_shard": alg_share,
    }
    eng.close()
    print(json.dumps(result))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--slab-width", type=int, default=0)
    ap.add_argument("--mode", default="auto", choices=["auto", "rowshard", "replica", "hybrid"],
                    help="multi-GPU leg whose rate is `value` (N > 1).  All three legs are always measured and printed "
                         "(`value_rowshard`, `value_hybrid`, `value_replica`): rowshard (BASELINE.json's layout: CSR rows + "
                         "embeddings sharded, one collective on the e4m3 iterate per sweep), hybrid (embeddings row-sharded, "
                         "one all-to-all of passage-score rows, PPR query-parallel on a replicated graph) and replica (queries "
                         "sharded, every GPU holds the whole index).  auto (default): rowshard when it is parity-green "
                         "(SURVEY 8(e): the primary figure), else hybrid, else replica (said in `value_leg`)")
    ap.add_argument("--cpu-budget-s", type=float, default=45.0)
    ap.add_argument("--cpu-queries", type=int, default=32, help="queries of the last batch checked against the CPU oracle")
    ap.add_argument("--cpu-ppr-queries", type=int, default=16,
                    help="... of which this many also get the PPR-only check (exact fp64 solve of the device's own reset vector)")
    ap.add_argument("--cpu-vec-queries", type=int, default=32, help="batch of the vectorised CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accel", action="store_true", help="skip the HRAG_OPT_ACCEL leg (never `value`)")
    ap.add_argument("--exchange-groups", type=int, default=2,
                    help="row-sharded mode: exchange groups pipelined against the sweeps")
    ap.add_argument("--sweep-launches", type=int, default=40)
    ap.add_argument("--no-rowshard", action="store_true", help="N > 1: skip the corpus-sharded legs (replica only)")
    ap.add_argument("--rowshard-timeout-s", type=float, default=600.0,
                    help="N > 1: abandon the corpus-sharded legs after this long (the line still prints what was measured)")
    ap.add_argument("--collective", default="allgather", choices=["allgather", "allreduce"],
                    help="N > 1, row-sharded leg: the per-sweep exchange of the e4m3 iterate -- in-place all-gather of the "
                         "owners' blocks (default), or BASELINE.json's literal all-reduce (foreign blocks zeroed, SUM over "
                         "the bytes: the same result at twice the wire bytes)")
    ap.add_argument("--shard-driver", default="python", choices=["python", "native"],
                    help="N > 1, row-sharded leg: the host loop around the hrag_shard_* steps in Python (dist.ShardedRetriever) or "
                         "inside the library (hrag_shard_retrieve with hrag_comm callbacks: dist.NativeShardedRetriever); "
                         "bit-identical results")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1 on configs[2]: skip the additional configs[3] figures (global batch 1024 on the same index)")
    ap.add_argument("--sell-sigma", type=int, default=0, help="hrag_opts.sell_sigma (SELL-C-sigma sorting window; 0 = global)")
    ap.add_argument("--engine-flags", type=int, default=0, help="hrag_opts.flags (HRAG_OPT_*), e.g. 2048 = XCD_BLOCKED")
    ap.add_argument("--locality", default="none", choices=["auto", "on", "degree", "none"],
                    help="HippoRAGEngine(locality=...), the engine's vertex numbering (the caller's ids are kept at the boundary); "
                         "none = as generated (the benchmark generator has no locality to find)")
    ap.add_argument("--ppr-tol", type=float, default=1.5e-6,
                    help="tolerance of the secondary leg that runs under the convergence contract (the headline runs "
                         "BASELINE.json's fixed 20 sweeps and reports the residual they leave)")
    ap.add_argument("--ppr-max-iters", type=int, default=29)
    args = ap.parse_args()
    args.locality = None if args.locality == "none" else args.locality

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not os.environ.get("HRAG_FORCE_DIST"):
        # the driver's command is plain `python bench.py --gpus N ...`: be our own launcher -- N ranks of this very
        # command, one per GPU, rendezvous on 127.0.0.1; rank 0's JSON line is the last line of our stdout
        from hipporag_amd.launch import self_spawn, visible_gpu_count
        n_dev = visible_gpu_count()      # no torch import: a box with too few GPUs must say so in seconds, not minutes
        if n_dev is not None and n_dev < args.gpus:
            sys.stderr.write(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs on this node (one rank per device); "
                             f"{n_dev} visible\n")
            return 2
        return self_spawn(args.gpus, os.path.abspath(__file__), sys.argv[1:])

    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1 or os.environ.get("HRAG_FORCE_DIST"):   # env: exercise the N>1 code on 1 GPU
        import bench_dist
        return bench_dist.bench_main(args, CONFIGS, rank, local_rank, world, roofline_fn=measure_roofline)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback on this path)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cfg = CONFIGS[args.config]
    V, E, D, B, seed = cfg["V"], cfg["E"], cfg["D"], args.batch or cfg["B"], cfg["seed"]
    if cfg.get("local_shards"):
        return bench_local_shards(args, cfg, dev)
    if cfg.get("shard_of"):
        return bench_shard_share(args, cfg, dev)

    t_setup = time.perf_counter()
    emb_dtype = torch.float16 if cfg.get("fp16") else torch.bfloat16
    if cfg.get("real2wiki"):
        from tools import real2wiki as rw
        kg = rw.build_kg(int(cfg["tiles"]))
        V, E = kg.num_vertices, kg.csr.nnz // 2
        pass_emb = torch.from_numpy(rw.mock_embeddings(kg.n_passages, seed + 1, D)).to(dev).to(emb_dtype)
        fact_emb = torch.from_numpy(rw.mock_embeddings(kg.n_facts, seed + 2, D)).to(dev).to(emb_dtype)
    else:
        kg = synth.make_kg(V, E, seed, power_law=bool(cfg.get("power_law")), community=int(cfg.get("community", 0)))
        if cfg.get("hash_order"):
            kg = synth.hash_order(kg, seed + 77)
        pass_emb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, dev, dtype=emb_dtype)
        fact_emb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, dev, dtype=emb_dtype)
    eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pass_emb, fact_emb, kg.subj_vertex, kg.obj_vertex,
                         kg.num_chunks, max_batch=B, max_topk=K_P, slab_width=args.slab_width, flags=args.engine_flags,
                         sell_sigma=args.sell_sigma, locality=args.locality)
    n_batches = args.steps + args.warmup
    qf = [synth.make_queries_torch(fact_emb, B, seed + 100 + i)[0] for i in range(n_batches)]
    qp = [synth.make_queries_torch(pass_emb, B, seed + 500 + i)[0] for i in range(n_batches)]
    cnt = torch.full((B,), K_F, dtype=torch.int32, device=dev)
    setup_s = time.perf_counter() - t_setup

    def step(i, tol=0.0, max_iters=0):
        idx, sc = eng.score_facts(qf[i], k=K_F)                         # phase A
        # identity "recognition memory" filter: all K_F candidates kept, device-side, no host sync
        return eng.retrieve(qp[i], idx, sc, cnt, link_top_k=K_F, damping=DAMPING,
                            passage_node_weight=PASSAGE_W, ppr_iters=PPR_ITERS, k=K_P, ppr_tol=tol,
                            ppr_max_iters=max_iters)

    def timed(tol=0.0, max_iters=0):
        for i in range(args.warmup):
            o = step(i, tol, max_iters)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()                    # no collector pass of the host interpreter inside the timed region
        t0 = time.perf_counter()
        res, used, flg = [], [], []
        for i in range(args.warmup, n_batches):
            o = step(i, tol, max_iters)
            res.append(o.residual); used.append(o.iters_used); flg.append(o.flags)     # device tensors: no sync
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gc.enable()
        # SURVEY 8(d)'s median of per-step HIP-event times, in a loop of its own AFTER the wall-clock region (an event
        # per step inside it cost sporadic 7-26 ms stalls on the small configurations); events on torch's current
        # stream = the stream the library launches on
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        marks[0].record()
        for i in range(args.warmup, n_batches):
            step(i, tol, max_iters)
            marks[i - args.warmup + 1].record()
        torch.cuda.synchronize()
        per_step = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
        from hipporag_amd._lib import PPR_ERR_FLOOR_F16, PPR_ERR_FLOOR_FP8, PPR_ERR_K
        res_max = float(torch.stack(res).max())
        floor = PPR_ERR_FLOOR_FP8 if B > 64 else PPR_ERR_FLOOR_F16
        contract = {"ppr_tol": tol, "ppr_max_iters": max(max_iters, PPR_ITERS) if tol > 0 else PPR_ITERS,
                    "ppr_residual_max": res_max,
                    # does what this leg leaves meet the mirror's DEFAULT tolerance (--ppr-tol, 1.5e-6)?  The fixed-count
                    # headline does not have to (BASELINE.json names a count, not a tolerance) -- this says what it costs
                    "meets_default_tol": bool(res_max <= (args.ppr_tol if args.ppr_tol > 0 else 1.5e-6)),
                    "default_tol": args.ppr_tol if args.ppr_tol > 0 else 1.5e-6,
                    "error_bound_from_residual": max(PPR_ERR_K * res_max, floor),
                    "error_bound_definition": "include/hrag.h: true relative error <= max(HRAG_PPR_ERR_K * residual, floor of the state type)",
                    "ppr_residual_definition": "damping / (1 - damping) * max over passages of the relative update of "
                                               "the passage score in the last sweep (include/hrag.h, hrag_retrieve)",
                    "sweeps_used_min": int(torch.stack(used).min()), "sweeps_used_max": int(torch.stack(used).max()),
                    "queries_flagged_not_converged": int((torch.stack(flg) & 16).ne(0).sum()),
                    "queries_flagged_fp8_saturated": int((torch.stack(flg) & 8).ne(0).sum()),
                    "step_ms_median_hip_events": per_step[len(per_step) // 2] if per_step else None,
                    "step_ms_min_max_hip_events": [per_step[0], per_step[-1]] if per_step else None}
        return o, el, contract

    out, elapsed, contract = timed()                                    # BASELINE.json: exactly 20 PPR iterations
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    qps = B * args.steps / elapsed
    if args.ppr_tol > 0:
        _, el_c, contract_c = timed(args.ppr_tol, args.ppr_max_iters)   # the mirror's default: the convergence contract
        contract_c.update({"value": B * args.steps / el_c, "unit": "queries/s", "ms_per_step": el_c * 1e3 / max(args.steps, 1)})
    else:
        contract_c = {"skipped": "--ppr-tol 0"}

    # Secondary leg, never `value`: HRAG_OPT_ACCEL -- the stages of the fp8-state PPR as Chebyshev steps (undirected
    # graph: real spectrum), fewer sweeps for the accuracy PPR_ITERS plain sweeps have; same K steps, same queries.
    # The headline above is BASELINE.json's literal 20 sweeps.
    accel, out_a = {"skipped": "--no-accel"}, None
    if not args.no_accel:
        from hipporag_amd._lib import OPT_ACCEL
        eng.set_flags(OPT_ACCEL, True)
        out_a, el_a, accel = timed()
        accel.update({"value": B * args.steps / el_a, "unit": "queries/s", "ms_per_step": el_a * 1e3 / max(args.steps, 1),
                      "what": "HRAG_OPT_ACCEL: ppr_iters = 20 names the accuracy, sweeps_used the sweeps that ran "
                              "(Chebyshev steps inside the stages of the fp8 / fp16 state; include/hrag.h).  Under a "
                              "tolerance only the fp8 state (batch > 64) accelerates"})
        same = out_a.doc_idx == out.doc_idx
        rel = ((out_a.doc_score - out.doc_score).abs() / out.doc_score.clamp_min(1e-30))[same]
        accel["vs_20_plain_sweeps_same_queries"] = {"top_k_positions_with_the_same_id": float(same.float().mean()),
                                                     "max_rel_score_diff_at_those": float(rel.max()) if rel.numel() else None}
        if args.ppr_tol > 0:
            _, el_ac, ac = timed(args.ppr_tol, args.ppr_max_iters)
            ac.update({"value": B * args.steps / el_ac, "unit": "queries/s", "ms_per_step": el_ac * 1e3 / max(args.steps, 1)})
            accel["with_convergence_contract"] = ac
        eng.set_flags(OPT_ACCEL, False)

    # phase breakdown (HIP events inside the library, same stream): five more steps on fresh queries, the one with the
    # median PPR time is reported (the last batch last: its results are what the oracle spot check reads)
    eng.set_profiling(True)
    runs = []
    for i in list(range(max(args.warmup, n_batches - 5), n_batches - 1)) + [n_batches - 1]:
        out = step(i)
        torch.cuda.synchronize()
        runs.append(eng.timings())
    phases = sorted(runs, key=lambda t: t["ppr_ms"])[len(runs) // 2]
    eng.set_profiling(False)

    roofline, f8, f16 = measure_roofline(eng, kg, V, B, phases, args.config, args.sweep_launches)
    nnz = kg.csr.nnz
    result = {
        "metric": "retrieval_queries_per_sec", "value": qps, "unit": "queries/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": cfg["label"], "V": V, "E": E, "nnz": nnz, "n_passages": kg.n_passages,
                   "n_facts": kg.n_facts, "dim": D, "global_batch": B, "ppr_iters": PPR_ITERS,
                   "linking_top_k": K_F, "retrieval_top_k": K_P, "damping": DAMPING,
                   "embedding_dtype": "fp16" if cfg.get("fp16") else "bf16",
                   "ppr_state_dtype": ("e4m3 staged corrections + fp32 true residual (fp32 arithmetic)" if f8 else
                                       "f16 hi + f16 correction (fp32 arithmetic)" if f16 else "f32"),
                   "sell_sigma": args.sell_sigma, "engine_flags": args.engine_flags, "locality": args.locality,
                   "locality_score_after_renumbering": eng.locality_score, "vertex_numbering": eng.numbering,
                   "engine_opt_flags": eng.opt_flags,
                   "parallelism": "1gpu"},
        "roofline": roofline,
        "step_ms_median_hip_events": contract.pop("step_ms_median_hip_events"),
        "step_ms_min_max_hip_events": contract.pop("step_ms_min_max_hip_events"),
        "ppr_contract": contract, "with_convergence_contract": contract_c, "with_accelerated_stages": accel,
        "phases_ms": {k: phases[k] for k in ("fact_sim_ms", "pass_sim_ms", "seed_ms", "ppr_ms", "rank_ms", "total_ms")},
        "sim_algorithmic_bytes": sim_algorithmic_bytes(kg.n_facts, kg.n_passages, D, B),
        # the similarity stage against ITS roofline -- max(HBM time, MFMA time) of the algorithmic
        # bytes / flops over the measured fact + passage phases (GEMMs + min/max + fused top-k)
        "similarity_roofline": (lambda sb, fl, ms: {
            "bound": "hbm" if sb / (HBM_PEAK_GBS * 1e9) >= fl / 2.5e15 else "mfma",
            "algorithmic_bytes": sb, "flops": fl, "phase_ms": ms,
            "achieved_gbs": sb / (ms * 1e-3) / 1e9, "achieved_tflops": fl / (ms * 1e-3) / 1e12,
            "frac": max(sb / (HBM_PEAK_GBS * 1e9), fl / 2.5e15) / (ms * 1e-3),
            "peaks": "8.0 TB/s HBM, 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)"})(
                sim_algorithmic_bytes(kg.n_facts, kg.n_passages, D, B), 2.0 * B * (kg.n_facts + kg.n_passages) * D,
                phases["fact_sim_ms"] + phases["pass_sim_ms"]),
        "n_long_rows": phases["n_long_rows"], "setup_s": setup_s,
        # hrag_engine_stats: device bytes of the shareable index / of this handle's workspace, PPR state types, the state
        # the last call ran on (8 = staged e4m3), why no e4m3 state if there is none
        "engine_stats": (lambda st: {k: st[k] for k in ("index_bytes", "workspace_bytes", "ppr_states", "last_ppr_state",
                                                         "fp8_unavailable", "fp8_unavailable_reasons")})(eng.stats()),
    }
    if not args.no_cpu_baseline:
        last = n_batches - 1
        also = ({"with_accelerated_stages": (out_a.doc_idx.cpu().numpy(), out_a.doc_score.cpu().numpy())}
                if out_a is not None else None)
        try:
            resets = device_reset_vectors(eng, kg, qf[last], qp[last], cnt, min(args.cpu_ppr_queries, B))
        except Exception as exc:        # the probe must never cost the line
            resets = None
            result["ppr_only_probe_error"] = f"{type(exc).__name__}: {exc}"
        cb, parity = cpu_baseline(kg, fact_emb, pass_emb, qf[last], qp[last], out.doc_idx.cpu().numpy(),
                                  out.doc_score.cpu().numpy(), args.cpu_budget_s, args.cpu_queries,
                                  vec_queries=args.cpu_vec_queries, also=also, device_resets=resets)
        result["cpu_baseline"] = cb
        result["parity_spot_check"] = parity
        result["speedup_vs_cpu_port"] = qps / cb["value"]
        if "value" in cb.get("vectorised", {}):
            result["speedup_vs_cpu_vectorised"] = qps / cb["vectorised"]["value"]
    eng.close()
    print(json.dumps(result))
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
