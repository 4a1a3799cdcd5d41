"""Multi-GPU modes of the retrieval hot path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) / gloo in the CPU tests.  The reference has no distributed code at all
(zero NCCL / MPI call sites; src/hipporag/HippoRAG.py:459 is a serial loop over the
queries); all three modes are MI355X-native additions.

rowshard  (the layout BASELINE.json's north star names; `value` of `bench.py --gpus N` when parity-green):
          the corpus is sharded row-wise -- rank g owns the CSR rows, the passage and the fact embedding rows of ITS
          equal-sized shard of a relabelled index (shard_index) -- and the e4m3 PPR iterate is replicated.
            * phase A : all-gather of each rank's local top-k fact candidates + MIN / MAX all-reduce (2 * B floats);
            * phase B : MIN / MAX all-reduce of the passage scores, MAX / SUM of the prior statistics; then per PPR
                        sweep every rank computes ITS rows of the new iterate and the owners' blocks are exchanged over
                        xGMI, one collective per exchange group, overlapped with the sweep of the other group(s).
                        TorchComm(collective="allgather") -- in-place all_gather_into_tensor of the owned blocks, 1 byte
                        per vertex and query -- or "allreduce": the north star's literal form (every rank zeroes the
                        rows it does not own, all-reduce SUM over the bytes: disjoint supports, so the sum IS the
                        gather; twice the wire bytes, kept for comparison);
            * final   : all-gather + merge of the local top-k lists.
          Every output row is produced by exactly one rank from the same replicated iterate: bit-identical to the
          single-GPU engine whatever the world size (tests/test_gpu_shard.py).
hybrid    embeddings row-sharded, ONE all-to-all of passage-score rows, PPR query-parallel on a replicated graph
          (HybridRetriever): 300x fewer wire bytes than the row-sharded PPR.
replica   queries are independent (the reference loop carries no cross-query state), every rank holds the whole
          index and serves its own slice of the batch with the single-GPU engine.  No data-path collective.

The compute of a rank is behind a small "stages" interface (hipporag_amd.engine.ShardStages = the hrag_shard_* C
entry points); tests drive the same orchestration with a CPU stand-in over gloo (tests/test_shard_orchestration.py,
tests/test_dist_gloo.py).
"""

from __future__ import annotations

import ctypes
import json
import os
import sys
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


def _td():
    import torch
    import torch.distributed as dist
    return torch, dist


def even_shards(n: int, world: int) -> List[Tuple[int, int]]:
    base, rem = divmod(n, world)
    out, lo = [], 0
    for g in range(world):
        hi = lo + base + (1 if g < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


# --------------------------------------------------------------------------------------------
# fp8-state row shards (hrag_shard_* entry points): equal-sized shards in an internal vertex order
# --------------------------------------------------------------------------------------------
@dataclass
class ShardedIndex:
    """The index relabelled so that shard g owns the vertex ids [g * rows_per_shard, (g + 1) * rows_per_shard):
    the passages of passage-embedding shard g (passage order), then entity vertices dealt round-robin by
    descending degree (every shard gets the same number of rows and ~the same number of matrix entries),
    then isolated padding vertices.  Passage POSITIONS (the ids the caller sees) are unchanged."""
    world: int
    rows_per_shard: int
    perm: np.ndarray                 # int64 [V_original]: original vertex id -> internal id
    csr: "object"                    # CSRGraph over world * rows_per_shard internal vertices
    passage_vertex: np.ndarray
    subj_vertex: Optional[np.ndarray]
    obj_vertex: Optional[np.ndarray]
    num_chunks: Optional[np.ndarray]
    passages: List[Tuple[int, int]]
    facts: List[Tuple[int, int]]

    @property
    def num_vertices(self) -> int:
        return self.world * self.rows_per_shard


def shard_index(csr, passage_vertex, world: int, subj_vertex=None, obj_vertex=None, num_chunks=None,
                n_facts: Optional[int] = None) -> ShardedIndex:
    from .graph import CSRGraph
    v = int(csr.num_vertices)
    pv = np.asarray(passage_vertex, dtype=np.int64)
    n_p = pv.shape[0]
    pshards = even_shards(n_p, world)
    deg = np.diff(np.asarray(csr.row_ptr, dtype=np.int64))
    is_pass = np.zeros(v, dtype=bool)
    is_pass[pv] = True
    ent = np.flatnonzero(~is_pass)
    ent = ent[np.argsort(-deg[ent], kind="stable")]               # heaviest first
    n_pass_of = np.array([hi - lo for lo, hi in pshards], dtype=np.int64)
    # 2 % of slack rows per shard: the balance below trades row counts for entry counts
    cap_ent = -(-ent.shape[0] // world) + max(1, ent.shape[0] // (50 * world))
    rps = int(n_pass_of.max()) + cap_ent
    perm = np.full(v, -1, dtype=np.int64)
    for g, (lo, hi) in enumerate(pshards):
        perm[pv[lo:hi]] = g * rps + np.arange(hi - lo)
    # longest-processing-time greedy: every entity (heaviest first) goes to the shard with the fewest matrix
    # entries so far that still has a free row; the remaining rows of a shard are isolated padding vertices
    import heapq
    load0 = [int(deg[pv[lo:hi]].sum()) for lo, hi in pshards]
    heap = [(load0[g], g) for g in range(world)]
    heapq.heapify(heap)
    count = [0] * world
    shard = np.empty(ent.shape[0], dtype=np.int64)
    local = np.empty(ent.shape[0], dtype=np.int64)
    ent_deg = deg[ent].tolist()
    for i, d in enumerate(ent_deg):
        ld, g = heapq.heappop(heap)
        shard[i], local[i] = g, count[g]
        count[g] += 1
        if count[g] < cap_ent:
            heapq.heappush(heap, (ld + d, g))
    perm[ent] = shard * rps + n_pass_of[shard] + local
    assert (perm >= 0).all() and np.unique(perm).shape[0] == v
    v_pad = world * rps
    rows = np.repeat(np.arange(v, dtype=np.int64), deg)
    new_r, new_c = perm[rows], perm[np.asarray(csr.col_idx, dtype=np.int64)]
    order = np.argsort(new_r * v_pad + new_c, kind="stable")
    row_ptr = np.zeros(v_pad + 1, dtype=np.int64)
    np.cumsum(np.bincount(new_r, minlength=v_pad), out=row_ptr[1:])
    col_sum = None
    if csr.col_sum is not None:
        col_sum = np.zeros(v_pad, dtype=np.float64)
        col_sum[perm] = csr.col_sum
    new_csr = CSRGraph(v_pad, row_ptr.astype(np.int32), new_c[order].astype(np.int32),
                       np.asarray(csr.val)[order], np.asarray(csr.raw)[order], col_sum)

    def relabel(a):
        if a is None:
            return None
        a = np.asarray(a, dtype=np.int64)
        return np.where(a >= 0, perm[np.clip(a, 0, v - 1)], -1).astype(np.int32)

    nc = None
    if num_chunks is not None:
        nc = np.zeros(v_pad, dtype=np.int32)
        nc[perm] = np.asarray(num_chunks, dtype=np.int32)
    nf = int(n_facts if n_facts is not None else (len(subj_vertex) if subj_vertex is not None else 0))
    return ShardedIndex(world, rps, perm, new_csr, perm[pv].astype(np.int32), relabel(subj_vertex),
                        relabel(obj_vertex), nc, pshards, even_shards(nf, world))


def build_shard_engine(sidx: ShardedIndex, pass_emb, fact_emb, rank: int, max_batch: int, max_topk: int,
                       flags: int = 0, sell_seg_len: int = 0):
    """The engine of shard `rank` of a ShardedIndex (full embedding matrices are sliced here)."""
    from .engine import HippoRAGEngine
    rps = sidx.rows_per_shard
    p_lo, p_hi = sidx.passages[rank]
    f_lo, f_hi = sidx.facts[rank]
    has_facts = fact_emb is not None and sidx.subj_vertex is not None
    return HippoRAGEngine(sidx.csr.rows(rank * rps, (rank + 1) * rps), sidx.passage_vertex, pass_emb[p_lo:p_hi],
                          fact_emb[f_lo:f_hi] if has_facts else None, sidx.subj_vertex if has_facts else None,
                          sidx.obj_vertex if has_facts else None, sidx.num_chunks if has_facts else None,
                          max_batch=max_batch, max_topk=max_topk, row_offset=rank * rps, passage_offset=p_lo,
                          fact_offset=f_lo, n_passages=len(sidx.passage_vertex),
                          n_facts=sidx.facts[-1][1] if has_facts else None, flags=flags, sell_seg_len=sell_seg_len)


class TorchComm:
    """The exchange steps of the row-sharded path over torch.distributed (nccl = RCCL over xGMI; gloo in
    the CPU tests).  State exchange = ONE in-place collective per exchange group: the owned rows of a group
    are one contiguous block at rank * own_bytes.  collective="allgather" (default): all_gather_into_tensor of the
    owned blocks; "allreduce": BASELINE.json's literal wording -- every rank zeroes the blocks it does not own and
    the group region is all-reduced (SUM over uint8: the supports are disjoint, so no byte ever adds to another
    and the sum is the gather, bit for bit) -- twice the wire bytes of the all-gather, kept for comparison."""

    def __init__(self, rank: int, world: int, group=None, collective: str = "allgather"):
        if collective not in ("allgather", "allreduce"):
            raise ValueError(f"collective must be 'allgather' or 'allreduce', not {collective!r}")
        self.rank, self.world, self.group, self.collective = rank, world, group, collective

    def all_reduce(self, t, op: str):
        torch, dist = _td()
        if self.world > 1:
            dist.all_reduce(t, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX,
                                   "sum": dist.ReduceOp.SUM}[op], group=self.group)
        return t

    def all_gather(self, t):
        torch, dist = _td()
        if self.world == 1:
            return [t]
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    def all_to_all(self, parts, shapes):
        """parts[d] goes to rank d; returns the list of what every rank sent here (shapes[s] = shape of rank s's
        part).  RCCL: one all_to_all; gloo (CPU tests) has none: paired isend / irecv."""
        torch, dist = _td()
        if self.world == 1:
            return [parts[0]]
        out = [torch.empty(tuple(shapes[s]), dtype=parts[0].dtype, device=parts[0].device) for s in range(self.world)]
        parts = [p.contiguous() for p in parts]
        if dist.get_backend(self.group) == "gloo":
            out[self.rank].copy_(parts[self.rank])
            reqs = []
            for d in range(self.world):
                if d != self.rank:
                    reqs.append(dist.isend(parts[d], self._global(d), group=self.group))
                    reqs.append(dist.irecv(out[d], self._global(d), group=self.group))
            for r in reqs:
                r.wait()
        else:
            dist.all_to_all(out, parts, group=self.group)
        return out

    def _global(self, g: int) -> int:
        torch, dist = _td()
        return g if self.group is None else dist.get_global_rank(self.group, g)

    def exchange(self, buf, lay, g: int):
        """Start the all-gather of exchange group g of state buffer `buf` (uint8 [state_bytes]); returns a
        handle for wait().  The collective is ordered after everything enqueued on the current stream."""
        torch, dist = _td()
        if self.world == 1:
            return None
        if lay.own_offset != self.rank * lay.own_bytes:
            raise ValueError("state exchange needs equal-sized row shards in rank order (dist.shard_index)")
        if self.world * lay.own_bytes + lay.slabs_per_group * 128 > lay.group_bytes:
            # the gathered region must end before the group's last row (row V, which stays zero: the target of the
            # masked-out gathers of the first sweep)
            raise ValueError("state exchange would overwrite the zero row: shards do not tile [0, V)")
        base = g * lay.group_bytes
        out = buf[base: base + self.world * lay.own_bytes]
        if self.collective == "allreduce":
            out[: lay.own_offset].zero_()
            out[lay.own_offset + lay.own_bytes:].zero_()
            return dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        inp = buf[base + lay.own_offset: base + lay.own_offset + lay.own_bytes]
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)

    def wait(self, handle):
        if handle is not None:
            handle.wait()          # the current stream waits for the collective; the host does not block


class HostStagedComm(TorchComm):
    """TorchComm for a process group whose backend takes HOST tensors only (gloo): every exchange step copies its
    operand to the host, runs the collective there and copies the result back -- same calls, same results, bytes over
    the host instead of xGMI.  What it is for: running the REAL hrag_shard_* kernels in several processes where RCCL
    has nothing to run on (two ranks sharing one GPU: tests/test_gpu_multi.py), or a node without a GPU fabric.  The
    copies synchronise the current stream, so the collective is ordered after the kernels that produced its operand
    and the next kernels after its result, like the device collectives of the base class."""

    def all_reduce(self, t, op: str):
        if self.world > 1 and t.is_cuda:
            h = t.cpu()
            super().all_reduce(h, op)
            t.copy_(h)
            return t
        return super().all_reduce(t, op)

    def all_gather(self, t):
        if self.world == 1 or not t.is_cuda:
            return super().all_gather(t)
        return [h.to(t.device) for h in super().all_gather(t.cpu())]

    def all_to_all(self, parts, shapes):
        if self.world == 1 or not parts[0].is_cuda:
            return super().all_to_all(parts, shapes)
        dev = parts[0].device
        return [h.to(dev) for h in super().all_to_all([p.cpu() for p in parts], shapes)]

    def exchange(self, buf, lay, g: int):
        if self.world == 1 or not buf.is_cuda:
            return super().exchange(buf, lay, g)
        base = g * lay.group_bytes
        region = buf[base: base + self.world * lay.own_bytes]
        h = region.cpu()                       # waits for the sweep that wrote the owned block
        if self.collective == "allreduce":     # the literal form: foreign blocks zeroed, SUM over the bytes
            h[: lay.own_offset].zero_()
            h[lay.own_offset + lay.own_bytes:].zero_()
        hb = _HostLayout(lay)
        super().exchange(h, hb, 0).wait()
        region.copy_(h)
        return None


class _HostLayout:
    """A shard layout whose group 0 starts at byte 0 of a host copy of ONE exchange group (HostStagedComm.exchange)."""

    def __init__(self, lay):
        self.own_offset, self.own_bytes = lay.own_offset, lay.own_bytes
        self.group_bytes, self.slabs_per_group = lay.group_bytes, lay.slabs_per_group


class LocalComm:
    """All `world` shards inside ONE process on ONE device, one thread per shard, meeting at barriers:
    the emulated gather SURVEY.md 8(e) asks for ("8 shards sequentially on one GPU must reproduce the
    single-GPU result").  The shards share the three state buffers (LocalComm.shared_buffers), so the
    state exchange is a barrier: every owner has enqueued its rows on the (common) stream before
    anybody enqueues the next sweep."""

    def __init__(self, rank: int, world: int, shared: dict):
        import threading
        self.rank, self.world, self.sh = rank, world, shared
        with shared.setdefault("_lock", threading.Lock()):
            if "_barrier" not in shared:
                shared["_barrier"] = threading.Barrier(world)
                shared["_slots"] = [None] * world

    def _barrier(self):
        self.sh["_barrier"].wait()

    def _collect(self, t):
        self.sh["_slots"][self.rank] = t
        self._barrier()
        parts = list(self.sh["_slots"])
        self._barrier()
        return parts

    def all_reduce(self, t, op: str):
        torch = _td()[0]
        parts = self._collect(t.clone())
        st = torch.stack(parts)
        # rank order: the same summation order on every shard (and in TorchComm's ring for world = 2)
        t.copy_({"min": lambda: st.min(0).values, "max": lambda: st.max(0).values, "sum": lambda: st.sum(0)}[op]())
        return t

    def all_gather(self, t):
        return self._collect(t)

    def all_to_all(self, parts, shapes):
        sent = self._collect(list(parts))            # sent[s][d] = what rank s sends to rank d
        return [sent[s][self.rank] for s in range(self.world)]

    def shared_buffers(self, key, make):
        if self.rank == 0:
            self.sh[key] = make()
        self._barrier()
        bufs = self.sh[key]
        self._barrier()
        return bufs

    def exchange(self, buf, lay, g: int):
        self._barrier()
        return None

    def wait(self, handle):
        pass


def merge_ranked(idx_parts, val_parts, k: int, topk):
    """Merge per-shard top-k lists (each sorted score desc, id desc; shard id ranges ascending in rank
    order) into the global top-k under the same rule.  Reversed and concatenated in rank order, "later
    position" == "larger (score, id)" among equal scores, so the library's positional tie rule
    reproduces the global order exactly.  Returns (ids int32 [B, k], values [B, k], -1 / 0 beyond)."""
    torch = _td()[0]
    cand_idx = torch.cat([t.flip(1) for t in idx_parts], dim=1).contiguous()
    cand_val = torch.cat([t.flip(1) for t in val_parts], dim=1)
    cand_val = torch.where(cand_idx < 0, torch.full_like(cand_val, float("-inf")), cand_val).contiguous()
    pos, top_val, _, _ = topk(cand_val, k)
    top_idx = torch.gather(cand_idx, 1, pos.clamp(min=0).long())
    top_idx = torch.where(pos < 0, torch.full_like(top_idx, -1), top_idx).to(torch.int32)
    top_val = torch.where(top_idx < 0, torch.zeros_like(top_val), top_val)
    return top_idx, top_val


class ShardedRetriever:
    """The hot path over fp8-state row shards: phase A / phase B with the exchange steps of
    include/hrag.h's hrag_shard_* section.  `stages` = the shard's engine (hipporag_amd.engine.ShardStages)
    or a CPU stand-in with the same methods (tests)."""

    def __init__(self, stages, comm, groups: int = 2):
        self.st, self.comm, self.groups = stages, comm, groups
        self._bufs = {}

    def _state(self, batch: int):
        if batch not in self._bufs:
            lay = self.st.shard_layout(batch, self.groups)
            make = lambda: [self.st.new_state(lay) for _ in range(3)]
            bufs = self.comm.shared_buffers(("state", batch), make) if hasattr(self.comm, "shared_buffers") else make()
            self._bufs[batch] = (lay, bufs)
        return self._bufs[batch]

    def score_facts(self, q_fact, k: int = 5):
        """Global (fact ids int32 [B, k], min-max normalised scores fp32 [B, k]), replicated."""
        torch = _td()[0]
        c = self.comm
        idx, val, mn, mx = self.st.shard_score_facts(q_fact, k)
        idx_all, val_all = c.all_gather(idx), c.all_gather(val)
        c.all_reduce(mn, "min")
        c.all_reduce(mx, "max")
        top_idx, top_val = merge_ranked(idx_all, val_all, k, self.st.topk)
        rng = (mx - mn).unsqueeze(1)
        norm = torch.where(rng == 0, torch.ones_like(top_val), (top_val - mn.unsqueeze(1)) / rng)   # misc_utils.py:130-139
        return top_idx, torch.where(top_idx < 0, torch.zeros_like(norm), norm)

    def retrieve(self, q_pass, kept_idx, kept_score, kept_count, *, link_top_k=5, damping=0.5,
                 passage_node_weight=0.05, ppr_iters=20, k=200, ppr_tol=0.0, ppr_max_iters=0,
                 check_saturation=True):
        """Returns (doc ids, doc scores, flags) -- and with ppr_tol > 0 additionally (residual, sweeps used): the
        convergence contract of hrag_retrieve on the shards.  Every shard measures the relative update of ITS
        passages; the measures are all-reduced (MAX) before every decision, so all shards run the same steps.
        check_saturation: the row shards have no wider state to repeat a batch on, so a batch that raised
        HRAG_FLAG_FP8_SATURATED (a violated scale bound: never on valid inputs) raises here instead of handing
        clipped scores on (include/hrag.h: "never return clipped scores"); it costs one host read per batch --
        a timing loop passes False and checks the flags afterwards."""
        st, c = self.st, self.comm
        b = q_pass.shape[0]
        lay, bufs = self._state(b)
        mn, mx = st.shard_passage_scores(q_pass)
        c.all_reduce(mn, "min")
        c.all_reduce(mx, "max")
        sv, sw, sc, flags = st.seeds(kept_idx, kept_score, kept_count, link_top_k)
        zmax, mass = st.shard_prior_stats(mn, mx, passage_node_weight, flags)
        c.all_reduce(zmax, "max")
        c.all_reduce(mass, "sum")
        contract = ppr_tol > 0
        if contract:
            n_steps = st.shard_ppr_begin(mn, mx, zmax, mass, passage_node_weight, (sv, sw, sc), flags, damping, ppr_iters,
                                         lay.n_groups, bufs, ppr_tol, ppr_max_iters)
        else:
            st.shard_ppr_begin(mn, mx, zmax, mass, passage_node_weight, (sv, sw, sc), flags, damping, ppr_iters,
                               lay.n_groups, bufs)
            n_steps = ppr_iters
        # group g's exchange overlaps with the sweep of the other groups: a sweep of group g only waits for
        # group g's previous exchange
        pend = [c.exchange(bufs[0], lay, g) for g in range(lay.n_groups)]
        est_global = False      # the engine's est already holds the all-reduced measure of the last final sweep
        for i in range(n_steps):
            ck = False
            for g in range(lay.n_groups):
                c.wait(pend[g])
                if contract:
                    xb, ck = st.shard_ppr_step(i, g)
                else:
                    xb = st.shard_ppr_sweep(i, g)
                pend[g] = c.exchange(bufs[xb], lay, g) if xb >= 0 else None
            est_global = False
            if ck:      # a final sweep that measured: the batch's residual over ALL passages, then the decision
                st.shard_ppr_est(c.all_reduce(st.shard_ppr_est(), "max"))
                st.shard_ppr_decide(i)
                est_global = True
                # the decision is the same on every shard (the measure was all-reduced).  Once it closes the gate of
                # the next step every later step is closed as well: stop issuing them -- each one would still cost a
                # full-state exchange per group for buffers nobody wrote (one host read per decision, <= 4 per batch)
                if not st.shard_ppr_gate_open(i + 1):
                    break
        for h in pend:          # an exchange started by the last step that ran
            c.wait(h)
        if contract:
            if not est_global:
                st.shard_ppr_est(c.all_reduce(st.shard_ppr_est(), "max"))
            idx, val, resid, used = st.shard_finish(mn, mx, flags, k, True)
        else:
            idx, val = st.shard_finish(mn, mx, flags, k)
        top_idx, top_val = merge_ranked(c.all_gather(idx), c.all_gather(val), k, st.topk)
        sat = (flags & 8).contiguous()               # raised on the shard that owns the row that saturated
        c.all_reduce(sat, "max")
        if check_saturation and bool(sat.any()):
            raise RuntimeError("fp8 PPR state saturated on a row shard (HRAG_FLAG_FP8_SATURATED): a static scale bound "
                               "was violated; the scores of the flagged queries are not trustworthy and the row-sharded "
                               "path has no wider state to repeat them on -- use the replica / hybrid mode for this batch")
        if contract:
            return top_idx, top_val, flags | sat, resid, used
        return top_idx, top_val, flags | sat


class NativeShardedRetriever:
    """ShardedRetriever with the host loop INSIDE the library (include/hrag.h: hrag_shard_score_facts_all /
    hrag_shard_retrieve, csrc/shard_driver.hip): one C call per phase; the library calls back only for the collectives
    (hrag_comm), which this class serves with the same `comm` object the Python loop uses (TorchComm over RCCL,
    HostStagedComm over gloo) -- so the two forms can be compared bit for bit (tests/test_gpu_multi.py).  A host without
    torch supplies four RCCL one-liners instead.  `stages` = ShardStages of this rank's shard engine."""

    def __init__(self, stages, comm, groups: int = 2):
        from . import _lib
        self.st, self.comm, self.groups = stages, comm, groups
        self._lib, self._L = stages.lib, _lib
        self._bufs, self._ws, self._pend = {}, {}, {}
        self._regions = []                 # device tensors the library may hand pointers into (workspace, state buffers)
        self._error = None
        L = _lib

        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as exc:       # a Python exception must not cross the C frame: report through the status
                    self._error = exc
                    return 1
            return run

        self._cb = (L.COMM_ALL_REDUCE(guard(self._all_reduce)), L.COMM_ALL_GATHER(guard(self._all_gather)),
                    L.COMM_EXCHANGE_BEGIN(guard(self._exchange_begin)), L.COMM_EXCHANGE_WAIT(guard(self._exchange_wait)))
        self._c = L.Comm(None, comm.rank, comm.world, *self._cb)

    # ---- device pointer -> view of a tensor this object owns
    def _view(self, ptr: int, nbytes: int, dtype):
        torch = _td()[0]
        for t in self._regions:
            off = ptr - t.data_ptr()
            if 0 <= off and off + nbytes <= t.numel() * t.element_size():
                return t.view(torch.uint8).reshape(-1)[off: off + nbytes].view(dtype)
        raise RuntimeError("hrag_comm callback: pointer outside the buffers handed to the library")

    def _all_reduce(self, user, buf, count, dtype, op, stream):
        torch = _td()[0]
        dt = {0: torch.float32, 1: torch.float64, 2: torch.int32}[dtype]
        self.comm.all_reduce(self._view(buf, count * (8 if dtype == 1 else 4), dt), {0: "min", 1: "max", 2: "sum"}[op])

    def _all_gather(self, user, send, recv, nbytes, stream):
        torch = _td()[0]
        parts = self.comm.all_gather(self._view(send, nbytes, torch.uint8))
        out = self._view(recv, nbytes * self.comm.world, torch.uint8)
        for r, p in enumerate(parts):
            out[r * nbytes: (r + 1) * nbytes].copy_(p)

    def _exchange_begin(self, user, region, own_bytes, group, stream):
        lay, bufs = self._cur
        for t in bufs:
            off = region - t.data_ptr()
            if 0 <= off < t.numel():
                if off != group * lay.group_bytes or own_bytes != lay.own_bytes:
                    raise RuntimeError("hrag_comm.exchange_begin: region does not match the shard layout")
                self._pend[group] = self.comm.exchange(t, lay, group)
                return
        raise RuntimeError("hrag_comm.exchange_begin: region outside the state buffers")

    def _exchange_wait(self, user, group, stream):
        self.comm.wait(self._pend.pop(group, None))

    # ---- buffers
    def _state(self, batch: int):
        if batch not in self._bufs:
            lay = self.st.shard_layout(batch, self.groups)
            make = lambda: [self.st.new_state(lay) for _ in range(3)]
            bufs = self.comm.shared_buffers(("state", batch), make) if hasattr(self.comm, "shared_buffers") else make()
            self._bufs[batch] = (lay, bufs)
        return self._bufs[batch]

    def _workspace(self, batch: int, k: int):
        torch = _td()[0]
        key = (batch, k)
        if key not in self._ws:
            n = int(self._lib.hrag_shard_workspace_bytes(self.st.h, self.comm.world, batch, k))
            self._ws[key] = torch.empty((n,), dtype=torch.uint8, device=self.st.device)
        return self._ws[key]

    def _call(self, fn, *args):
        from ._lib import check
        self._error = None
        st = fn(*args)
        if self._error is not None:
            raise self._error
        check(st)

    def score_facts(self, q_fact, k: int = 5):
        torch = _td()[0]
        from .engine import _stream
        q = self.st.e._q(q_fact)
        b = q.shape[0]
        ws = self._workspace(b, k)
        idx = torch.empty((b, k), dtype=torch.int32, device=self.st.device)
        val = torch.empty((b, k), dtype=torch.float32, device=self.st.device)
        self._regions = [ws]
        self._call(self._lib.hrag_shard_score_facts_all, self.st.h, ctypes.byref(self._c), q.data_ptr(), b, k, ws.data_ptr(),
                   ws.numel(), idx.data_ptr(), val.data_ptr(), _stream())
        return idx, val

    def retrieve(self, q_pass, kept_idx, kept_score, kept_count, *, link_top_k=5, damping=0.5, passage_node_weight=0.05,
                 ppr_iters=20, k=200, ppr_tol=0.0, ppr_max_iters=0, check_saturation=True):
        """Same contract as ShardedRetriever.retrieve: (doc ids, doc scores, flags[, residual, sweeps used])."""
        torch = _td()[0]
        from .engine import _stream
        q = self.st.e._q(q_pass)
        b = q.shape[0]
        lay, bufs = self._state(b)
        ws = self._workspace(b, k)
        dev = self.st.device
        kept_idx = kept_idx.to(dev, torch.int32).contiguous()
        kept_score = kept_score.to(dev, torch.float32).contiguous()
        kept_count = kept_count.to(dev, torch.int32).contiguous()
        idx = torch.empty((b, k), dtype=torch.int32, device=dev)
        val = torch.empty((b, k), dtype=torch.float32, device=dev)
        flags = torch.empty((b,), dtype=torch.int32, device=dev)
        contract = ppr_tol > 0
        resid = torch.empty((b,), dtype=torch.float32, device=dev) if contract else None
        used = torch.empty((b,), dtype=torch.int32, device=dev) if contract else None
        self._regions, self._cur, self._pend = [ws] + list(bufs), (lay, bufs), {}
        self.st.e._p8_batch = b
        self._call(self._lib.hrag_shard_retrieve, self.st.h, ctypes.byref(self._c), q.data_ptr(), b, kept_idx.data_ptr(),
                   kept_score.data_ptr(), kept_count.data_ptr(), kept_idx.shape[1], link_top_k, damping, passage_node_weight,
                   ppr_iters, max(ppr_max_iters, ppr_iters), ppr_tol, k, lay.n_groups, bufs[0].data_ptr(),
                   bufs[1].data_ptr(), bufs[2].data_ptr(), ws.data_ptr(), ws.numel(), idx.data_ptr(), val.data_ptr(),
                   flags.data_ptr(), resid.data_ptr() if contract else None, used.data_ptr() if contract else None, _stream())
        if check_saturation and bool((flags & 8).any()):
            raise RuntimeError("fp8 PPR state saturated on a row shard (HRAG_FLAG_FP8_SATURATED): a static scale bound "
                               "was violated; use the replica / hybrid mode for this batch")
        return (idx, val, flags, resid, used) if contract else (idx, val, flags)


class HybridRetriever:
    """The hybrid multi-GPU mode SURVEY.md 8(e) ends on: the EMBEDDINGS are row-sharded over the GPUs (the part of
    the index that grows with the corpus: configs[4] holds 20 GB of them), the PPR runs QUERY-PARALLEL on a replicated
    graph (164 MB of CSR at configs[3]) with no exchange at all.

      phase A   every rank scores ALL queries of the global batch against ITS fact rows (local top-k) -> all-gather of
                the candidates + MIN / MAX all-reduce -> the same merge as the row-sharded mode (replicated result);
      phase B   every rank scores ALL queries against ITS passage rows [B, Np / N] -> ONE all-to-all hands rank r the
                score rows of ITS B / N queries over all passages [B / N, Np] -> hrag_retrieve_scored on those queries.

    Wire per global batch: B * Np * 4 bytes in total for the all-to-all ((N - 1) / N of it crosses links) plus the
    candidate lists -- 0.45 GB at configs[3] (B = 1024) against 20 sweeps x 0.91 GB RECEIVED PER GPU of the
    row-sharded PPR (146 GB in total): a factor 300.  Every passage score is the same MFMA chain whatever matrix slice
    it is computed from, so the result is bit-identical to the single-GPU engine on the same queries.
    sim = ShardStages of this rank's shard engine (dist.build_shard_engine: its embedding shards are what is used),
    ppr = a HippoRAGEngine WITHOUT embeddings over the whole graph, pshards = the passage ranges of the ranks."""

    def __init__(self, sim, ppr, comm, pshards):
        self.sim, self.ppr, self.comm, self.pshards = sim, ppr, comm, list(pshards)

    def score_facts(self, q_fact, k: int = 5):
        torch = _td()[0]
        c = self.comm
        idx, val, mn, mx = self.sim.shard_score_facts(q_fact, k)
        idx_all, val_all = c.all_gather(idx), c.all_gather(val)
        c.all_reduce(mn, "min")
        c.all_reduce(mx, "max")
        top_idx, top_val = merge_ranked(idx_all, val_all, k, self.sim.topk)
        rng = (mx - mn).unsqueeze(1)
        norm = torch.where(rng == 0, torch.ones_like(top_val), (top_val - mn.unsqueeze(1)) / rng)   # misc_utils.py:130-139
        return top_idx, torch.where(top_idx < 0, torch.zeros_like(norm), norm)

    def my_rows(self, batch: int):
        w, r = self.comm.world, self.comm.rank
        if batch % w:
            raise ValueError(f"the global batch ({batch}) must be a multiple of the world size ({w})")
        return slice(r * (batch // w), (r + 1) * (batch // w))

    def retrieve(self, q_pass, kept_idx, kept_score, kept_count, **kw):
        """q_pass / kept_*: the GLOBAL batch (replicated); returns this rank's RetrieveOutput for its B / N queries."""
        torch = _td()[0]
        c, w = self.comm, self.comm.world
        b = q_pass.shape[0]
        mine = self.my_rows(b)
        bn = b // w
        s_local = self.sim.e.sim_scores("passages", q_pass)                     # [B, Np_local]
        parts = [s_local[d * bn:(d + 1) * bn] for d in range(w)]
        got = c.all_to_all(parts, [(bn, hi - lo) for lo, hi in self.pshards])
        scores = torch.cat(got, dim=1).contiguous()                            # [B / N, Np] in passage order
        return self.ppr.retrieve_scored(scores, kept_idx[mine], kept_score[mine], kept_count[mine], **kw)


def run_local_shards(world: int, sidx: "ShardedIndex", pass_emb, fact_emb, q_fact, q_pass, retrieve_kw: dict,
                     groups: int, device, max_topk: int, filter_fn=None, timings: Optional[dict] = None,
                     sell_seg_len: int = 0, native: bool = False):
    """All `world` shards of `sidx` as threads of THIS process on ONE device, meeting at barriers (LocalComm) and
    sharing the three e4m3 state buffers: the emulated gather SURVEY.md 8(e) prescribes.  Returns rank 0's
    (fact idx, fact score, doc idx, doc score, flags) as numpy arrays after checking that every rank computed the
    same replicated result.  Used by tests/test_gpu_shard.py and by `bench.py --config cfg4local` (the parity-checked
    form of BASELINE configs[3] when only one GPU is at hand).  native: the host loop inside the library
    (NativeShardedRetriever: hrag_shard_retrieve calling back into LocalComm from every shard thread)."""
    import threading
    torch = _td()[0]
    from .engine import ShardStages
    shared, results, errors = {}, [None] * world, []
    b = q_fact.shape[0]

    def worker(rank):
        try:
            torch.cuda.set_device(device)
            eng = build_shard_engine(sidx, pass_emb, fact_emb, rank, max_batch=b, max_topk=max_topk,
                                     sell_seg_len=sell_seg_len)
            rs = (NativeShardedRetriever if native else ShardedRetriever)(ShardStages(eng), LocalComm(rank, world, shared),
                                                                          groups=groups)
            torch.cuda.synchronize()
            shared["_barrier"].wait()
            t0 = time.perf_counter()
            idx, sc = rs.score_facts(q_fact, k=5)
            cnt = torch.full((b,), 5, dtype=torch.int32, device=device)
            if filter_fn is not None:
                idx, sc, cnt = filter_fn(idx, sc)
            d_idx, d_sc, flags = rs.retrieve(q_pass, idx, sc, cnt, **retrieve_kw)[:3]   # ppr_tol > 0: + (residual, sweeps)
            torch.cuda.synchronize()
            if timings is not None and rank == 0:
                timings["wall_s_all_shards_on_one_device"] = time.perf_counter() - t0
            results[rank] = tuple(t.cpu().numpy() for t in (idx, sc, d_idx, d_sc, flags))
            shared["_barrier"].wait()          # nobody frees its engine while another shard still runs
            eng.close()
        except Exception as exc:               # a dead shard must not leave the others at a barrier for ever
            errors.append((rank, exc))
            try:
                shared["_barrier"].abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=1200)
    if errors:
        raise RuntimeError(f"shard threads failed: {errors}")
    for r in range(1, world):
        for a, w in zip(results[r], results[0]):
            np.testing.assert_array_equal(a, w)
    return results[0]


def build_ppr_engine(graph, passage_vertex, subj_vertex, obj_vertex, num_chunks, dim: int, max_batch: int,
                     max_topk: int, flags: int = 0):
    """The PPR side of the hybrid mode: the whole graph + the fact lookup arrays, NO embeddings."""
    from .engine import HippoRAGEngine
    return HippoRAGEngine(graph, passage_vertex, None, None, subj_vertex, obj_vertex, num_chunks, dim=dim,
                          max_batch=max_batch, max_topk=max_topk, flags=flags)


def run_local_hybrid(world: int, kg_arrays: dict, sidx: "ShardedIndex", pass_emb, fact_emb, q_fact, q_pass,
                     retrieve_kw: dict, device, max_topk: int, timings: Optional[dict] = None):
    """The hybrid mode with all `world` ranks as threads of this process on ONE device (LocalComm).  kg_arrays: the
    ORIGINAL index (csr, passage_vertex, subj_vertex, obj_vertex, num_chunks) for the PPR engines.  Returns, in
    query order, (fact idx, fact score) of the global batch and the concatenated per-rank (doc idx, doc score, flags)."""
    import threading
    torch = _td()[0]
    from .engine import ShardStages
    shared, results, errors = {}, [None] * world, []
    b = q_fact.shape[0]
    bn = b // world

    def worker(rank):
        try:
            torch.cuda.set_device(device)
            sim = build_shard_engine(sidx, pass_emb, fact_emb, rank, max_batch=b, max_topk=max_topk)
            ppr = build_ppr_engine(kg_arrays["csr"], kg_arrays["passage_vertex"], kg_arrays["subj_vertex"],
                                   kg_arrays["obj_vertex"], kg_arrays["num_chunks"], sim.dim, max_batch=bn,
                                   max_topk=max_topk)
            hy = HybridRetriever(ShardStages(sim), ppr, LocalComm(rank, world, shared), sidx.passages)
            torch.cuda.synchronize()
            shared["_barrier"].wait()
            t0 = time.perf_counter()
            idx, sc = hy.score_facts(q_fact, k=5)
            cnt = torch.full((b,), 5, dtype=torch.int32, device=device)
            out = hy.retrieve(q_pass, idx, sc, cnt, **retrieve_kw)
            torch.cuda.synchronize()
            if timings is not None and rank == 0:
                timings["wall_s_all_ranks_on_one_device"] = time.perf_counter() - t0
            results[rank] = tuple(t.cpu().numpy() for t in (idx, sc, out.doc_idx, out.doc_score, out.flags))
            shared["_barrier"].wait()
            sim.close()
            ppr.close()
        except Exception as exc:
            errors.append((rank, exc))
            try:
                shared["_barrier"].abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=1200)
    if errors:
        raise RuntimeError(f"hybrid threads failed: {errors}")
    for r in range(1, world):
        np.testing.assert_array_equal(results[r][0], results[0][0])     # the merged fact candidates are replicated
        np.testing.assert_array_equal(results[r][1], results[0][1])
    return (results[0][0], results[0][1], np.concatenate([r[2] for r in results]),
            np.concatenate([r[3] for r in results]), np.concatenate([r[4] for r in results]))
