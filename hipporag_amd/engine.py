"""Torch-facing wrapper of the libhrag.so engine (include/hrag.h).

PyTorch supplies device memory and the current HIP stream; all arithmetic on the path happens in
the hand-written gfx950 kernels behind the C ABI.  There is no CPU or eager-torch fallback: without
a GPU (or without libhrag.so) constructing an engine raises.
"""

from __future__ import annotations

import ctypes as C
import logging
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import (EmbedDesc, FactDesc, GraphDesc, Opts, ShardLayout, Timings, check, FLAG_DPR_FALLBACK,
                   FLAG_FP8_SATURATED, FLAG_ZERO_MASS, FLAG_ZERO_PHRASE, SEED_STRIDE)
from .graph import CSRGraph


def _torch():
    import torch
    return torch


def _ptr(a) -> int:
    """Address of a numpy array (host) or torch tensor (host or device)."""
    if a is None:
        return 0
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data
    if not a.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return a.data_ptr()


def _stream() -> int:
    return _torch().cuda.current_stream().cuda_stream


class SplitRows:
    """Embedding rows already in an HRAG_F32_SPLIT engine's layout (fp16 [rows, 3 * dim] on the device): what
    gather_embeddings returns on such an engine, accepted by the constructor (HRAG_F32_SPLIT_ROWS)."""

    def __init__(self, tensor, dim: int):
        self.tensor, self.dim = tensor, int(dim)
        self.shape = (tensor.shape[0], self.dim)

    def __getitem__(self, rows):        # row slices (dist.build_shard_engine)
        return SplitRows(self.tensor[rows], self.dim)


def _as_16bit(emb):
    """Accept a torch bf16 / fp16 tensor, a numpy float16 array, uint16 bf16 bit patterns (numpy / torch), or fp32
    (numpy / torch: the fp32-faithful HRAG_F32_SPLIT engine); return (obj, rows, dim, dtype) with dtype 0 = bf16,
    1 = fp16, 2 = fp32 split (hrag_dtype)."""
    torch = _torch()
    if isinstance(emb, SplitRows):
        return emb.tensor.contiguous(), emb.tensor.shape[0], emb.dim, 3
    if isinstance(emb, np.ndarray):
        if emb.dtype == np.float32:
            emb = np.ascontiguousarray(emb)
            return emb, emb.shape[0], emb.shape[1], 2
        if emb.dtype == np.float16:
            emb = np.ascontiguousarray(emb)
            return emb, emb.shape[0], emb.shape[1], 1
        if emb.dtype != np.uint16:
            raise TypeError("numpy embeddings must be float32 (fp32-faithful engine), float16, or uint16 bf16 bit "
                            "patterns (hipporag_amd.graph.float_to_bf16_bits)")
        emb = np.ascontiguousarray(emb)
        return emb, emb.shape[0], emb.shape[1], 0
    if emb.dtype not in (torch.bfloat16, torch.float16, torch.uint16, torch.int16, torch.float32):
        raise TypeError("tensor embeddings must be torch.float32, torch.bfloat16 or torch.float16 (or raw bf16 bit patterns)")
    emb = emb.contiguous()
    return emb, emb.shape[0], emb.shape[1], 2 if emb.dtype == torch.float32 else 1 if emb.dtype == torch.float16 else 0


_TORCH_DTYPE_OF = {0: "bfloat16", 1: "float16", 2: "float32", 3: "float32"}


@dataclass
class RetrieveOutput:
    doc_idx: "object"     # torch int32 [B, k] passage positions, best first
    doc_score: "object"   # torch fp32  [B, k]
    flags: "object"       # torch int32 [B]
    residual: "object" = None    # torch fp32 [B]: damping / (1 - damping) * the relative size of the last sweep's update
                                 # of the passage scores (include/hrag.h, hrag_retrieve); -1 where not measured
    iters_used: "object" = None  # torch int32 [B]: PPR sweeps that ran
    all_scores: "object" = None  # torch fp32 [B, Np] (retrieve_converged(want_all_scores=True)): every passage's score


logger = logging.getLogger(__name__)


def host_copy_async(t):
    """(host tensor, event): the copy of device tensor `t` to pinned host memory, enqueued on the current stream; the
    event (None for a tensor that already lives on the host) completes when the copy has.  Waiting for the event does
    not wait for anything enqueued after it -- what a pipeline over batches needs (t.cpu() drains the stream)."""
    torch = _torch()
    if not t.is_cuda:
        return t, None
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    with torch.cuda.device(t.device):                 # the copy and its event on the tensor's own device and current stream
        h.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
    return h, ev


def host_wait(pair):
    """The numpy view of a host_copy_async() result, once its copy has completed."""
    h, ev = pair
    if ev is not None:
        ev.synchronize()
    return h.numpy()


class PendingRetrieve:
    """A batch that HippoRAGEngine.retrieve_converged_start() enqueued.  finish() is the host half of the convergence
    contract (see retrieve_converged); `repeated` tells whether anything was run again (the output tensors then
    differ from what was enqueued first)."""

    def __init__(self, eng, run, out, flags_h, damping, ppr_iters, ppr_tol, ppr_max_iters, want_all_scores):
        self.eng, self.run, self.out, self.flags_h = eng, run, out, flags_h
        self.damping, self.ppr_iters, self.ppr_tol, self.ppr_max_iters = damping, ppr_iters, ppr_tol, ppr_max_iters
        self.want_all_scores = want_all_scores
        self.repeated, self.flags = False, None

    def finish(self) -> RetrieveOutput:
        torch = _torch()
        from ._lib import FLAG_FP8_SATURATED, FLAG_NOT_CONVERGED, OPT_NO_F16, OPT_NO_FP8
        eng, run, out = self.eng, self.run, self.out
        damping, ppr_iters, ppr_tol, ppr_max_iters = self.damping, self.ppr_iters, self.ppr_tol, self.ppr_max_iters

        def on_wider_state(fn, bits=OPT_NO_FP8):
            had = eng.opt_flags & bits               # restore what the engine was created with
            eng.set_flags(bits, True)
            try:
                return fn()
            finally:
                if bits & ~had:
                    eng.set_flags(bits & ~had, False)

        flags = host_wait(self.flags_h).copy()
        if (flags & FLAG_FP8_SATURATED).any() and (eng.opt_flags & _lib.OPT_ACCEL):
            # accelerated stages (HRAG_OPT_ACCEL): a violated scale bound falls back to the plain plan first
            logger.warning("fp8 PPR state saturated under HRAG_OPT_ACCEL for %d queries: repeating the batch on the plain plan",
                           int(((flags & FLAG_FP8_SATURATED) != 0).sum()))
            eng.set_flags(_lib.OPT_ACCEL, False)
            try:
                out = run()
            finally:
                eng.set_flags(_lib.OPT_ACCEL, True)
            self.repeated = True
            flags = out.flags.cpu().numpy()
        if (flags & FLAG_FP8_SATURATED).any():
            logger.warning("fp8 PPR state saturated for %d queries: repeating the batch on the wider state",
                           int(((flags & FLAG_FP8_SATURATED) != 0).sum()))
            out = on_wider_state(run)
            self.repeated = True
            flags = out.flags.cpu().numpy()
        self.out, self.flags = out, flags            # flags: the final flag words on the host (numpy)
        if ppr_tol <= 0 or not (flags & FLAG_NOT_CONVERGED).any():
            return out
        self.repeated = True
        resid, used = out.residual.cpu().numpy(), out.iters_used.cpu().numpy()
        for _ in range(6):
            rows = np.flatnonzero(flags & FLAG_NOT_CONVERGED)
            if not len(rows):
                break
            done = int(used[rows].max())
            need = done + int(np.ceil(np.log(max(float(resid[rows].max()) / ppr_tol, 1.0)) / -np.log(damping))) + 4
            need = min(max(need, done + 4), max(ppr_max_iters, ppr_iters))
            if need <= done:
                break                                   # ppr_max_iters reached: the flag stays
            logger.info("PPR not converged for %d queries after %d sweeps (residual %.2g > %.2g): repeating them "
                        "with %d sweeps", len(rows), done, float(resid[rows].max()), ppr_tol, need)
            rt = torch.as_tensor(rows, device=eng.device)
            # fp32 state: the scores that converge last are orders of magnitude below the largest one
            o2 = on_wider_state(lambda: run(rt, need, need), OPT_NO_FP8 | OPT_NO_F16)
            out.doc_idx[rt], out.doc_score[rt] = o2.doc_idx, o2.doc_score
            out.flags[rt], out.residual[rt], out.iters_used[rt] = o2.flags, o2.residual, o2.iters_used
            if self.want_all_scores:
                out.all_scores[rt] = o2.all_scores
            flags[rows], resid[rows], used[rows] = (o2.flags.cpu().numpy(), o2.residual.cpu().numpy(),
                                                    o2.iters_used.cpu().numpy())
        left = (flags & FLAG_NOT_CONVERGED) != 0
        if left.any():
            logger.warning("PPR residual above ppr_tol=%.2g for %d queries after ppr_max_iters=%d sweeps (max %.2g)",
                           ppr_tol, int(left.sum()), ppr_max_iters, float(resid[left].max()))
        return out


def _graph_fingerprint(g) -> tuple:
    """A cheap identity of a CSR graph's arrays (sizes + strided samples of row_ptr / col_idx / val, no full pass): equal for
    the arrays an engine was created from, different -- with overwhelming likelihood -- once they were edited in place."""
    def sample(a):
        a = np.asarray(a)
        if a.size == 0:
            return b""
        step = max(1, a.size // 1024)
        return np.ascontiguousarray(a.reshape(-1)[::step][:1024]).tobytes()
    return (int(g.num_vertices), int(np.asarray(g.col_idx).shape[0]), hash(sample(g.row_ptr)), hash(sample(g.col_idx)),
            hash(sample(g.val)))


class HippoRAGEngine:
    """Device-resident retrieval state: CSR graph, bf16 fact / passage embeddings, lookup arrays.

    Mirrors what ``HippoRAG.prepare_retrieval_objects`` stages on the host (reference
    src/hipporag/HippoRAG.py:1287-1389).  Row-sharded construction (multi-GPU): pass the row shard
    of the CSR (``CSRGraph.rows``) with ``row_offset`` and the embedding shards with their offsets.
    """

    def __init__(self, graph: CSRGraph, passage_vertex, passage_emb, fact_emb=None,
                 subj_vertex=None, obj_vertex=None, num_chunks=None, *, max_batch: int = 256,
                 max_topk: int = 200, slab_width: int = 0, long_row_nnz: int = 0,
                 row_offset: int = 0, passage_offset: int = 0, fact_offset: int = 0,
                 n_passages: Optional[int] = None, n_facts: Optional[int] = None,
                 device: Optional[int] = None, flags: int = 0, segment_nnz: int = 0, sell_seg_len: int = 0,
                 dim: Optional[int] = None, sell_sigma: int = 0, locality: Optional[str] = None):
        """locality ("auto" / "on" / "degree" / None): the graph compiler's vertex numbering; everything the caller sees --
        passage positions, fact ids, hrag_ppr's vertex order -- keeps the caller's numbering.  "on": graph.locality_order
        (vertices by the first passage that links them) + SELL-C-sigma windows + the XCD-blocked launch; "auto": that
        numbering, the two switches only when the renumbered matrix actually has locality (graph.locality_score >= 0.3);
        "degree": graph.degree_order (hubs first) -- an experiment switch, not a default: -4 ... -12 % per call for
        B <= 8 and +1.2 % for the wide batch on the 1M-vertex benchmark graph, but -6 % on the 100k-vertex one
        (docs/experiments/README.md).  Unsharded engines only.

        passage_emb=None (with dim=...): an engine WITHOUT embeddings -- the PPR side of the hybrid multi-GPU mode
        (dist.HybridRetriever): passage scores arrive through retrieve_scored(), seeds still come from subj_vertex /
        obj_vertex / num_chunks."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("HippoRAGEngine needs an MI355X-class GPU: no HIP device is visible "
                               "(there is no CPU fallback on this path)")
        self._lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._handle = C.c_void_p(0)

        pv = np.ascontiguousarray(passage_vertex, dtype=np.int32)
        caller_graph = graph            # what the lazy undirected-graph test looks at (a relabelling changes nothing for it)
        self._perm = self._inv_perm = None
        self.locality_score = self.numbering = None
        if locality:
            if locality not in ("auto", "on", "degree"):
                raise ValueError("locality must be None, 'auto', 'on' or 'degree'")
            if row_offset != 0 or graph.row_ptr.shape[0] - 1 != graph.num_vertices:
                raise ValueError("the locality numbering is for unsharded engines (dist.shard_index relabels shards)")
            from .graph import degree_order, locality_order, locality_score, relabel_csr
            windows = False
            if locality == "degree":
                perm = degree_order(graph, pv)
            else:
                perm = locality_order(graph, pv)
                self.locality_score = locality_score(graph, perm=perm)       # of the matrix as it will be renumbered
                windows = locality == "on" or self.locality_score >= 0.3
            self.numbering = "degree" if locality == "degree" else "locality"
            graph = relabel_csr(graph, perm)
            pv = np.ascontiguousarray(perm[pv], dtype=np.int32)
            if subj_vertex is not None:
                sv_ = np.asarray(subj_vertex, dtype=np.int64)
                ov_ = np.asarray(obj_vertex, dtype=np.int64)
                subj_vertex = np.where(sv_ >= 0, perm[np.clip(sv_, 0, None)], -1).astype(np.int32)
                obj_vertex = np.where(ov_ >= 0, perm[np.clip(ov_, 0, None)], -1).astype(np.int32)
                nc_ = np.zeros(graph.num_vertices, dtype=np.int32)
                nc_[perm] = np.asarray(num_chunks, dtype=np.int32)
                num_chunks = nc_
            if windows:
                sell_sigma = sell_sigma or 16384
                flags |= _lib.OPT_XCD_BLOCKED
            self._perm = torch.from_numpy(perm).to(self.device)              # caller's vertex id -> engine's
            inv = np.empty_like(perm)
            inv[perm] = np.arange(perm.shape[0])
            self._inv_perm = torch.from_numpy(inv).to(self.device)
        self.n_passages = int(pv.shape[0]) if n_passages is None else int(n_passages)
        if passage_emb is None:
            if not dim or fact_emb is not None:
                raise ValueError("an engine without passage embeddings needs dim= and takes no fact embeddings")
            p_obj, p_rows, dt = None, self.n_passages, 0
        else:
            p_obj, p_rows, dim, dt = _as_16bit(passage_emb)
        self.emb_dtype = getattr(torch, _TORCH_DTYPE_OF[dt])    # the dtype queries travel in (fp32 on a split engine)
        self.f32_split = dt in (2, 3)
        row_ptr = np.ascontiguousarray(graph.row_ptr, dtype=np.int32)
        col_idx = np.ascontiguousarray(graph.col_idx, dtype=np.int32)
        val = np.ascontiguousarray(graph.val, dtype=np.float32)
        n_rows = row_ptr.shape[0] - 1
        # weighted degrees (optional): enable the fp8-state PPR for batches > 64 (csrc/ppr8.hip)
        col_sum = getattr(graph, "col_sum", None)
        if col_sum is not None:
            col_sum = np.ascontiguousarray(col_sum, dtype=np.float64)
            if col_sum.shape[0] != graph.num_vertices:
                raise ValueError("col_sum must have one entry per vertex")
        gd = GraphDesc(graph.num_vertices, row_offset, n_rows, col_idx.shape[0], _ptr(row_ptr),
                       _ptr(col_idx), _ptr(val), self.n_passages, _ptr(pv),
                       _ptr(col_sum) if col_sum is not None else None)
        pd = EmbedDesc(p_rows, passage_offset, dim, dt, _ptr(p_obj) if p_obj is not None else None)
        fdesc = fd = None
        keep = [pv, p_obj, row_ptr, col_idx, val, col_sum]
        self.n_facts = 0
        if fact_emb is None and passage_emb is None and subj_vertex is not None:
            # scores-only engine: the fact lookup arrays (seeds) without fact embeddings
            sv = np.ascontiguousarray(subj_vertex, dtype=np.int32)
            ov = np.ascontiguousarray(obj_vertex, dtype=np.int32)
            nc = np.ascontiguousarray(num_chunks, dtype=np.int32)
            self.n_facts = int(sv.shape[0])
            fdesc = EmbedDesc(0, 0, dim, dt, None)
            fd = FactDesc(self.n_facts, _ptr(sv), _ptr(ov), _ptr(nc))
            keep += [sv, ov, nc]
        if fact_emb is not None:
            f_obj, f_rows, f_dim, f_dt = _as_16bit(fact_emb)
            if f_dim != dim or (f_dt != dt and not (f_dt in (2, 3) and dt in (2, 3))):
                raise ValueError("fact / passage embedding dims or dtypes differ")
            sv = np.ascontiguousarray(subj_vertex, dtype=np.int32)
            ov = np.ascontiguousarray(obj_vertex, dtype=np.int32)
            nc = np.ascontiguousarray(num_chunks, dtype=np.int32)
            if nc.shape[0] != graph.num_vertices:
                raise ValueError("num_chunks must have one entry per vertex")
            self.n_facts = int(sv.shape[0]) if n_facts is None else int(n_facts)
            if sv.shape[0] != self.n_facts or ov.shape[0] != self.n_facts:
                raise ValueError("subj_vertex / obj_vertex must cover all (global) facts")
            fdesc = EmbedDesc(f_rows, fact_offset, dim, f_dt, _ptr(f_obj))
            fd = FactDesc(self.n_facts, _ptr(sv), _ptr(ov), _ptr(nc))
            keep += [f_obj, sv, ov, nc]
        # HRAG_OPT_ACCEL needs an undirected graph (real spectrum of the sweep operator): checked where the graph is at
        # hand (row sums of the adjacency == its column sums); a directed graph keeps the plain plan, loudly
        # -- evaluated only when the flag is asked for (here, or at the first set_flags(OPT_ACCEL)): the test walks the
        # whole matrix, which a plain engine (and every incremental re-prepare of one) has no use for
        import weakref
        self._undirected = None if (row_offset == 0 and n_rows == graph.num_vertices) else False
        self._graph_print = _graph_fingerprint(caller_graph)      # what was uploaded: the lazy test must see the same arrays
        try:
            self._graph_ref = weakref.ref(caller_graph)
        except TypeError:       # an object that cannot be weakly referenced: hold it only until the test has run
            self._graph_ref = lambda g=caller_graph: g
        if (flags & _lib.OPT_ACCEL) and not self._is_undirected():
            logger.warning("HRAG_OPT_ACCEL dropped: the graph does not look undirected (or has no col_sum / is a row shard)")
            flags &= ~_lib.OPT_ACCEL
        self.opt_flags = int(flags)      # HRAG_OPT_* bits as set_flags leaves them
        opts = Opts(max_batch, max_topk, slab_width, long_row_nnz, self.device.index, flags, segment_nnz, sell_seg_len, sell_sigma)
        with torch.cuda.device(self.device):
            check(self._lib.hrag_engine_create(C.byref(gd), C.byref(fdesc) if fdesc else None,
                                               C.byref(pd), C.byref(fd) if fd else None,
                                               C.byref(opts), C.byref(self._handle)))
        del keep
        self.num_vertices = int(graph.num_vertices)
        self.row_offset, self.n_rows = int(row_offset), int(n_rows)
        self.passage_rows, self.passage_offset = int(p_rows), int(passage_offset)
        self.fact_rows = int(fact_emb.shape[0]) if fact_emb is not None else 0
        self.fact_offset = int(fact_offset)
        self.dim = int(dim)
        self.max_batch, self.max_topk = int(max_batch), int(max_topk)

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        """Destroy the handle.  An engine closes the workspaces it handed out first (they borrow its index buffers)."""
        for w in list(getattr(self, "_workspaces", ())):
            w.close()
        if getattr(self, "_handle", None) and self._handle.value:
            check(self._lib.hrag_engine_destroy(self._handle))
            self._handle = C.c_void_p(0)
            parent = getattr(self, "_parent", None)
            if parent is not None and self in parent._workspaces:
                parent._workspaces.remove(self)

    def workspace(self) -> "HippoRAGEngine":
        """A second WORKSPACE on this index (hrag_workspace_create): an engine object with the same methods
        that shares this engine's device-resident index -- graph, SELL-8 matrices, embeddings: nothing is copied -- and owns
        its own per-call buffers, so another thread can run score_facts / retrieve on it, on its own stream, WHILE this
        engine (or another workspace) is inside a call.  Results are bit-identical.  Close workspaces before the engine
        (close() of the engine does it); gather_embeddings() on any handle needs all handles idle."""
        import copy
        root = getattr(self, "_parent", None) or self
        w = copy.copy(root)                      # plain attributes: sizes, numbering, dtype, the library
        w._handle = C.c_void_p(0)
        w._workspaces = []
        w._parent = root
        with _torch().cuda.device(root.device):
            check(root._lib.hrag_workspace_create(root._handle, C.byref(w._handle)))
        if not hasattr(root, "_workspaces"):
            root._workspaces = []
        root._workspaces.append(w)
        return w

    def stats(self) -> dict:
        """hrag_engine_stats: device bytes of the shared index / of this handle's workspace, the PPR state types the
        handle can run (and why not the e4m3 one), the state the last retrieve ran on, call counters."""
        from ._lib import FP8_UNAVAILABLE, Stats
        s = Stats()
        check(self._lib.hrag_engine_stats(self._handle, C.byref(s)))
        d = {name: int(getattr(s, name)) for name, _ in Stats._fields_ if name != "reserved"}
        d["fp8_unavailable_reasons"] = [txt for bit, txt in FP8_UNAVAILABLE.items() if s.fp8_unavailable & bit]
        return d

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ helpers
    def _q(self, q):
        torch = _torch()
        if q.dtype != self.emb_dtype:      # queries travel in the engine's embedding dtype (bf16 / fp16)
            q = q.to(self.emb_dtype)
        q = q.to(self.device).contiguous()
        if q.dim() != 2 or q.shape[1] != self.dim:
            raise ValueError(f"queries must be [B, {self.dim}]")
        return q

    def _empty(self, shape, dtype):
        return _torch().empty(shape, dtype=dtype, device=self.device)

    def set_profiling(self, on: bool):
        check(self._lib.hrag_set_profiling(self._handle, 1 if on else 0))

    def timings(self) -> dict:
        t = Timings()
        check(self._lib.hrag_get_timings(self._handle, C.byref(t)))
        return {name: getattr(t, name) for name, _ in Timings._fields_}

    def layout(self, batch: int) -> Tuple[int, int]:
        bc, ns = C.c_int32(), C.c_int32()
        check(self._lib.hrag_ppr_layout(self._handle, batch, C.byref(bc), C.byref(ns)))
        return bc.value, ns.value

    # ------------------------------------------------------------------ fused entry points
    def score_facts(self, q_fact, k: int = 5):
        """Phase A: (fact ids int32 [B,k], min-max normalised scores fp32 [B,k])."""
        torch = _torch()
        q = self._q(q_fact)
        b = q.shape[0]
        idx = self._empty((b, k), torch.int32)
        sc = self._empty((b, k), torch.float32)
        check(self._lib.hrag_score_facts(self._handle, q.data_ptr(), b, k, idx.data_ptr(), sc.data_ptr(),
                                         _stream()))
        return idx, sc

    def retrieve(self, q_pass, kept_idx, kept_score, kept_count, *, link_top_k: int = 5,
                 damping: float = 0.5, passage_node_weight: float = 0.05, ppr_iters: int = 20,
                 k: int = 200, ppr_tol: float = 0.0, ppr_max_iters: int = 0,
                 want_residual: bool = True) -> RetrieveOutput:
        """Phase B: seeds + passage prior + PPR + ranking (DPR ranking where kept_count == 0).
        ppr_tol / ppr_max_iters: the convergence contract of include/hrag.h (0 = exactly ppr_iters sweeps);
        the output carries the per-query residual and the sweeps that ran."""
        torch = _torch()
        q = self._q(q_pass)
        b = q.shape[0]
        kept_idx = kept_idx.to(self.device, torch.int32).contiguous()
        kept_score = kept_score.to(self.device, torch.float32).contiguous()
        kept_count = kept_count.to(self.device, torch.int32).contiguous()
        kf = kept_idx.shape[1]
        if kept_idx.shape != (b, kf) or kept_score.shape != (b, kf) or kept_count.shape != (b,):
            raise ValueError("kept_idx / kept_score must be [B, kf], kept_count [B]")
        idx = self._empty((b, k), torch.int32)
        sc = self._empty((b, k), torch.float32)
        flags = self._empty((b,), torch.int32)
        resid = self._empty((b,), torch.float32) if want_residual or ppr_tol > 0 else None
        used = self._empty((b,), torch.int32) if resid is not None else None
        check(self._lib.hrag_retrieve(self._handle, q.data_ptr(), b, kept_idx.data_ptr(),
                                      kept_score.data_ptr(), kept_count.data_ptr(), kf, link_top_k,
                                      damping, passage_node_weight, ppr_iters, max(ppr_max_iters, ppr_iters),
                                      ppr_tol, k, idx.data_ptr(), sc.data_ptr(), flags.data_ptr(),
                                      resid.data_ptr() if resid is not None else None,
                                      used.data_ptr() if used is not None else None, _stream()))
        return RetrieveOutput(idx, sc, flags, resid, used)

    def retrieve_scored(self, pass_scores, kept_idx, kept_score, kept_count, *, link_top_k: int = 5,
                        damping: float = 0.5, passage_node_weight: float = 0.05, ppr_iters: int = 20, k: int = 200,
                        ppr_tol: float = 0.0, ppr_max_iters: int = 0) -> RetrieveOutput:
        """retrieve() with the raw passage scores fp32 [B, >= Np] supplied by the caller (hrag_retrieve_scored)."""
        torch = _torch()
        sc_in = pass_scores.to(self.device, torch.float32)
        if sc_in.stride(-1) != 1:
            sc_in = sc_in.contiguous()
        b = sc_in.shape[0]
        kept_idx = kept_idx.to(self.device, torch.int32).contiguous()
        kept_score = kept_score.to(self.device, torch.float32).contiguous()
        kept_count = kept_count.to(self.device, torch.int32).contiguous()
        kf = kept_idx.shape[1]
        idx, sc = self._empty((b, k), torch.int32), self._empty((b, k), torch.float32)
        flags, resid, used = self._empty((b,), torch.int32), self._empty((b,), torch.float32), self._empty((b,), torch.int32)
        check(self._lib.hrag_retrieve_scored(self._handle, sc_in.data_ptr(), sc_in.stride(0), b, kept_idx.data_ptr(),
                                             kept_score.data_ptr(), kept_count.data_ptr(), kf, link_top_k, damping,
                                             passage_node_weight, ppr_iters, max(ppr_max_iters, ppr_iters), ppr_tol, k,
                                             idx.data_ptr(), sc.data_ptr(), flags.data_ptr(), resid.data_ptr(),
                                             used.data_ptr(), _stream()))
        return RetrieveOutput(idx, sc, flags, resid, used)

    def retrieve_converged(self, q_pass, kept_idx, kept_score, kept_count, *, damping: float = 0.5,
                           ppr_iters: int = 20, ppr_tol: float = 1.5e-6, ppr_max_iters: int = 400,
                           want_all_scores: bool = False, **kw) -> RetrieveOutput:
        """retrieve() + what the host owes the convergence contract (include/hrag.h): a batch whose fp8 state
        saturated is repeated on the wider state (never clipped scores); queries the engine flags
        HRAG_FLAG_NOT_CONVERGED -- its sweep budget did not reach ppr_tol: a slowly mixing graph -- are repeated,
        those queries only, on the wider state with the sweeps their residual asks for (it contracts by `damping`
        per sweep), up to ppr_max_iters.  The reference's PRPACK does the same thing implicitly: it iterates to
        1e-10 whatever the graph (HippoRAG.py:1736-1743).  A flag that survives means ppr_max_iters was too small.

        = retrieve_converged_start(...).finish(); callers that have host work to do while the batch runs (the mirror's
        batch pipeline, retriever.iter_batched_retrieve) use the two halves."""
        return self.retrieve_converged_start(q_pass, kept_idx, kept_score, kept_count, damping=damping, ppr_iters=ppr_iters,
                                             ppr_tol=ppr_tol, ppr_max_iters=ppr_max_iters,
                                             want_all_scores=want_all_scores, **kw).finish()

    def retrieve_converged_start(self, q_pass, kept_idx, kept_score, kept_count, *, damping: float = 0.5,
                                 ppr_iters: int = 20, ppr_tol: float = 1.5e-6, ppr_max_iters: int = 400,
                                 want_all_scores: bool = False, **kw) -> "PendingRetrieve":
        """Enqueue the batch and the copy of its flag words to (pinned) host memory; nothing waits.  .finish() waits for
        the flags -- not for whatever was enqueued behind them -- and does the host half of the contract."""
        q = self._q(q_pass)
        kept_idx, kept_score, kept_count = (t.to(self.device) for t in (kept_idx, kept_score, kept_count))

        def run(rows=None, iters=ppr_iters, max_iters=ppr_max_iters):
            sel = slice(None) if rows is None else rows
            o = self.retrieve(q[sel], kept_idx[sel], kept_score[sel], kept_count[sel], damping=damping,
                              ppr_iters=iters, ppr_tol=ppr_tol, ppr_max_iters=max(max_iters, iters), **kw)
            if want_all_scores:     # every passage's score (callers that want more than max_topk documents)
                o.all_scores = self.last_doc_scores(o.doc_idx.shape[0])
            return o

        out = run()
        return PendingRetrieve(self, run, out, host_copy_async(out.flags), damping, ppr_iters, ppr_tol, ppr_max_iters,
                               want_all_scores)

    def last_doc_scores(self, batch: int):
        """fp32 [batch, Np]: the scores of ALL passages behind the last retrieve() / retrieve_scored() (hrag_last_doc_scores)."""
        torch = _torch()
        out = self._empty((batch, self.passage_rows), torch.float32)
        check(self._lib.hrag_last_doc_scores(self._handle, batch, out.data_ptr(), self.passage_rows, _stream()))
        return out

    def dense_retrieve(self, q_pass, k: int = 200):
        torch = _torch()
        q = self._q(q_pass)
        b = q.shape[0]
        idx = self._empty((b, k), torch.int32)
        sc = self._empty((b, k), torch.float32)
        check(self._lib.hrag_dense_retrieve(self._handle, q.data_ptr(), b, k, idx.data_ptr(),
                                            sc.data_ptr(), _stream()))
        return idx, sc

    # ------------------------------------------------------------------ seams
    def sim_scores(self, which: str, q):
        """Raw cosine scores fp32 [B, rows] against "facts" or "passages" (np.dot seam)."""
        torch = _torch()
        qq = self._q(q)
        rows = self.fact_rows if which == "facts" else self.passage_rows
        out = self._empty((qq.shape[0], rows), torch.float32)
        check(self._lib.hrag_sim_scores(self._handle, 0 if which == "facts" else 1, qq.data_ptr(),
                                        qq.shape[0], out.data_ptr(), _stream()))
        return out

    def ppr(self, reset, damping: float = 0.5, iters: int = 20):
        """run_ppr core for B reset vectors fp32 [B, V] -> (x fp32 [B, V], flags int32 [B])."""
        torch = _torch()
        r = reset.to(self.device, torch.float32).contiguous()
        if r.dim() != 2 or r.shape[1] != self.num_vertices:
            raise ValueError(f"reset must be [B, {self.num_vertices}]")
        b = r.shape[0]
        if self._perm is not None:      # the engine's vertex numbering: reset_engine[:, new] = reset[:, old]
            r = r.index_select(1, self._inv_perm).contiguous()
        x = self._empty((b, self.num_vertices), torch.float32)
        flags = self._empty((b,), torch.int32)
        check(self._lib.hrag_ppr(self._handle, r.data_ptr(), b, damping, iters, x.data_ptr(),
                                 flags.data_ptr(), _stream()))
        if self._perm is not None:
            x = x.index_select(1, self._perm).contiguous()
        return x, flags

    def ppr_sweeps(self, batch: int, n: int, damping: float = 0.5, main_only: bool = False,
                   f16: bool = False, small: bool = False, f8: bool = False, f8_mode: str = "C", f8_rio: int = 0,
                   f8_gather_replay: bool = False):
        """Measurement hook: n sweeps of the fp32 slab kernel, of the fp16-state kernel (f16=True), of
        the small-batch kernel (small=True, batch <= 8) or of the fp8-state kernel (f8=True; f8_mode picks
        the kernel instantiation: "C" stage sweep, "B" boundary, "B0" first boundary, "F" final;
        f8_gather_replay=True: only the state-row gathers of a stage sweep, nothing computed or written)."""
        flags = (1 if main_only else 0) | (2 if f16 else 0) | (4 if small else 0) | (8 if f8 else 0)
        flags |= ({"C": 0, "B": 1, "F": 2, "B0": 3}[f8_mode] << 4) | ((f8_rio & 3) << 6)
        flags |= 256 if (f8 and f8_gather_replay) else 0
        check(self._lib.hrag_ppr_sweeps(self._handle, batch, n, damping, flags, _stream()))

    def _is_undirected(self) -> bool:
        """graph.looks_undirected of the engine's graph, evaluated once, on first use (needs the graph object the engine
        was created from to be alive still: the mirror and the adapter keep theirs)."""
        if self._undirected is None:
            from .graph import looks_undirected
            g = self._graph_ref() if getattr(self, "_graph_ref", None) else None
            if g is None:
                raise ValueError("HRAG_OPT_ACCEL was not asked for at creation and the graph object is gone: the "
                                 "undirected-graph test cannot run -- pass flags=OPT_ACCEL to HippoRAGEngine()")
            if _graph_fingerprint(g) != self._graph_print:
                raise ValueError("the graph object was modified after the engine was created from it: the undirected-graph "
                                 "test would look at another matrix than the one on the device -- create the engine with "
                                 "flags=OPT_ACCEL, or keep the graph arrays unchanged until set_flags(OPT_ACCEL)")
            self._undirected = bool(looks_undirected(g))
            self._graph_ref = None
        return self._undirected

    def set_flags(self, flags: int, on: bool = True):
        """Set / clear HRAG_OPT_* bits after creation (e.g. _lib.OPT_NO_FP8 to rerun a saturated batch).
        OPT_ACCEL, first time: runs graph.looks_undirected on the graph OBJECT the engine was created from -- it must
        still be alive and unchanged (a fingerprint of its arrays is checked); engines that will need the flag later can
        pass it at creation instead, which runs the test there."""
        if on and (flags & _lib.OPT_ACCEL) and not self._is_undirected():
            raise ValueError("HRAG_OPT_ACCEL needs an undirected graph with col_sum on an unsharded engine "
                             "(hipporag_amd.graph.looks_undirected): this engine's graph does not qualify")
        check(self._lib.hrag_engine_set_flags(self._handle, flags, 1 if on else 0))
        self.opt_flags = (self.opt_flags | flags) if on else (self.opt_flags & ~flags)

    def gather_embeddings(self, which: str, src_rows, new_rows=None):
        """The embedding matrix of the next engine after an index update, composed ON THE DEVICE:
        out[i] = this engine's row src_rows[i] (>= 0) or new_rows[-src_rows[i] - 1].  new_rows: bf16 / fp16
        tensor or uint16 bit patterns (host or device).  Returns a device tensor [n, dim] in the engine's dtype."""
        torch = _torch()
        src = torch.as_tensor(np.ascontiguousarray(src_rows, dtype=np.int32)).to(self.device)
        n = int(src.shape[0])
        fresh = None
        if new_rows is not None and len(new_rows):
            obj, rows, dim, dt = _as_16bit(new_rows)
            if dim != self.dim or getattr(torch, _TORCH_DTYPE_OF[dt]) != self.emb_dtype or dt == 3:
                raise ValueError("new rows must match the engine's embedding dim / dtype")
            if self.f32_split:     # fp32 rows -> the engine's [hi | lo | hi] layout, on the device
                x = (torch.from_numpy(obj) if isinstance(obj, np.ndarray) else obj).to(self.device).contiguous()
                fresh = torch.empty((rows, 3 * self.dim), dtype=torch.int16, device=self.device)
                check(self._lib.hrag_split_f32(x.data_ptr(), rows, self.dim, 0, 0, fresh.data_ptr(), _stream()))
            else:
                fresh = (torch.from_numpy(obj.view(np.int16)) if isinstance(obj, np.ndarray) else obj.view(torch.int16)).to(self.device).contiguous()
        if n and int(src.min().item()) < 0 and (fresh is None or int((-src.min()).item()) > fresh.shape[0]):
            raise ValueError("src_rows refers to a new row that was not given")
        # (a split engine's rows are 3 * dim fp16 elements; the result is only ever handed to the next engine)
        out = (torch.empty((n, 3 * self.dim), dtype=torch.float16, device=self.device) if self.f32_split
               else torch.empty((n, self.dim), dtype=self.emb_dtype, device=self.device))
        if which not in ("facts", "passages"):
            raise ValueError("which must be 'facts' or 'passages'")
        held = self.fact_rows if which == "facts" else self.passage_rows
        if n and int(src.max().item()) >= held:
            raise ValueError(f"src_rows refers to row {int(src.max().item())} of {held} held {which} rows")
        check(self._lib.hrag_engine_gather_embeddings(self._handle, 0 if which == "facts" else 1, src.data_ptr(), n,
                                                      fresh.data_ptr() if fresh is not None else None,
                                                      out.data_ptr(), _stream()))
        return SplitRows(out, self.dim) if self.f32_split else out

    # ------------------------------------------------------------------ row shard (include/hrag.h hrag_shard_*)
    def shard_layout(self, batch: int, groups: int = 0) -> ShardLayout:
        lay = ShardLayout()
        check(self._lib.hrag_shard_layout_query(self._handle, batch, groups, C.byref(lay)))
        return lay

    def shard_score_facts(self, q_fact, k: int = 5):
        """Local phase A: (global fact ids [B,k], RAW scores [B,k], local min [B], local max [B])."""
        torch = _torch()
        q = self._q(q_fact)
        b = q.shape[0]
        idx = self._empty((b, k), torch.int32)
        val = self._empty((b, k), torch.float32)
        mn = self._empty((b,), torch.float32)
        mx = self._empty((b,), torch.float32)
        check(self._lib.hrag_shard_score_facts(self._handle, q.data_ptr(), b, k, idx.data_ptr(), val.data_ptr(),
                                               mn.data_ptr(), mx.data_ptr(), _stream()))
        return idx, val, mn, mx

    def shard_passage_scores(self, q_pass):
        torch = _torch()
        q = self._q(q_pass)
        b = q.shape[0]
        mn = self._empty((b,), torch.float32)
        mx = self._empty((b,), torch.float32)
        check(self._lib.hrag_shard_passage_scores(self._handle, q.data_ptr(), b, mn.data_ptr(), mx.data_ptr(),
                                                  _stream()))
        return mn, mx

    def shard_prior_stats(self, mn, mx, passage_node_weight: float, flags):
        torch = _torch()
        b = mn.shape[0]
        zmax = self._empty((b,), torch.float32)
        mass = self._empty((2 * b,), torch.float64)
        check(self._lib.hrag_shard_prior_stats(self._handle, mn.data_ptr(), mx.data_ptr(), passage_node_weight,
                                               flags.data_ptr(), b, zmax.data_ptr(), mass.data_ptr(), _stream()))
        return zmax, mass

    def shard_ppr_begin(self, mn, mx, zmax, mass, passage_node_weight, seeds, flags, damping, ppr_iters,
                        n_groups, bufs, ppr_tol: float = 0.0, ppr_max_iters: int = 0) -> int:
        """Returns the number of steps of the session (ppr_iters, plus the conditional steps of the convergence
        contract when ppr_tol > 0: run them all, the device decides which ones do anything)."""
        sv, sw, sc = seeds
        b = mn.shape[0]
        self._p8_batch = b
        n_steps = C.c_int32(0)
        check(self._lib.hrag_shard_ppr_begin(self._handle, mn.data_ptr(), mx.data_ptr(), zmax.data_ptr(),
                                             mass.data_ptr(), passage_node_weight, sv.data_ptr(), sw.data_ptr(),
                                             sc.data_ptr(), flags.data_ptr(), b, damping, ppr_iters,
                                             max(ppr_max_iters, ppr_iters), ppr_tol, n_groups,
                                             bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(),
                                             C.byref(n_steps), _stream()))
        return n_steps.value

    def shard_ppr_step(self, step: int, group: int):
        """(state buffer to exchange or -1, True when the step is a final sweep that measures the residual: the convergence
        contract's all-reduce + decision follow it)."""
        x, ck = C.c_int32(-1), C.c_int32(0)
        check(self._lib.hrag_shard_ppr_sweep(self._handle, step, group, C.byref(x), C.byref(ck), _stream()))
        return x.value, bool(ck.value)

    def shard_ppr_sweep(self, sweep: int, group: int) -> int:
        return self.shard_ppr_step(sweep, group)[0]

    def shard_ppr_est(self, est=None):
        """est=None: this shard's measure of the latest final sweep (fp32 [B]); est given: write the all-reduced values
        back."""
        torch = _torch()
        b = self._p8_batch
        if est is None:
            out = self._empty((b,), torch.float32)
            check(self._lib.hrag_shard_ppr_est(self._handle, out.data_ptr(), 0, _stream()))
            return out
        check(self._lib.hrag_shard_ppr_est(self._handle, est.data_ptr(), 1, _stream()))
        return est

    def shard_ppr_decide(self, step: int):
        check(self._lib.hrag_shard_ppr_decide(self._handle, step, _stream()))

    def shard_ppr_gate_open(self, step: int) -> bool:
        """Will step `step` do anything (hrag_shard_ppr_gate; synchronises the stream)?"""
        o = C.c_int32(0)
        check(self._lib.hrag_shard_ppr_gate(self._handle, step, C.byref(o), _stream()))
        return bool(o.value)

    def shard_finish(self, mn, mx, flags, k: int, want_residual: bool = False):
        torch = _torch()
        b = mn.shape[0]
        idx = self._empty((b, k), torch.int32)
        val = self._empty((b, k), torch.float32)
        resid = self._empty((b,), torch.float32) if want_residual else None
        used = self._empty((b,), torch.int32) if want_residual else None
        check(self._lib.hrag_shard_finish(self._handle, mn.data_ptr(), mx.data_ptr(), flags.data_ptr(), b, k,
                                          idx.data_ptr(), val.data_ptr(),
                                          resid.data_ptr() if want_residual else None,
                                          used.data_ptr() if want_residual else None, _stream()))
        return (idx, val, resid, used) if want_residual else (idx, val)


class CapturedPipeline:
    """Phase A + identity filter + phase B of one fixed batch size captured into a HIP graph.

    The C ABI only enqueues kernels (no allocation, no synchronisation), so the ~35 launches of a
    single-query retrieval (``retrieve_ircot`` issues one per reasoning step, HippoRAG.py:526,539)
    collapse into one graph launch.  ``__call__`` copies the new queries into the graph's static input
    buffers, replays, and returns views of the static outputs (valid until the next call):

        pipe = CapturedPipeline(engine, batch=1, k=200)
        fact_idx, fact_score, doc_idx, doc_score, flags = pipe(q_fact, q_pass)

    A host-side LLM filter cannot sit inside a graph: use score_facts / retrieve directly for that.
    """

    def __init__(self, engine: "HippoRAGEngine", batch: int, *, k_f: int = 5, k: int = 200, **retrieve_kw):
        torch = _torch()
        self.engine, self.batch = engine, int(batch)
        dev, dt = engine.device, engine.emb_dtype
        self._qf = torch.zeros((batch, engine.dim), dtype=dt, device=dev)
        self._qp = torch.zeros((batch, engine.dim), dtype=dt, device=dev)
        self._cnt = torch.full((batch,), k_f, dtype=torch.int32, device=dev)

        def run():
            idx, sc = engine.score_facts(self._qf, k=k_f)
            out = engine.retrieve(self._qp, idx, sc, self._cnt, link_top_k=k_f, k=k, **retrieve_kw)
            return idx, sc, out.doc_idx, out.doc_score, out.flags

        with torch.cuda.device(dev):
            run()                                    # warm-up (lazy module loads) outside the capture
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._out = run()

    def __call__(self, q_fact, q_pass):
        self._qf.copy_(q_fact)
        self._qp.copy_(q_pass)
        self.graph.replay()
        return self._out


def topk_rows(scores, k: int, *, n: Optional[int] = None, idx_offset: int = 0, normalize: bool = False,
              want_minmax: bool = False):
    """Row-wise top-k of a device fp32 matrix with the library ranking rule (hrag_topk_rows)."""
    torch = _torch()
    lib = _lib.load()
    if not scores.is_cuda:
        raise RuntimeError("topk_rows runs on the GPU only")
    s = scores.to(torch.float32).contiguous()
    b, ld = s.shape
    n = ld if n is None else n
    idx = torch.empty((b, k), dtype=torch.int32, device=s.device)
    val = torch.empty((b, k), dtype=torch.float32, device=s.device)
    mn = torch.empty((b,), dtype=torch.float32, device=s.device)
    mx = torch.empty((b,), dtype=torch.float32, device=s.device)
    check(lib.hrag_topk_rows(s.data_ptr(), b, n, ld, k, idx_offset, 1 if normalize else 0, idx.data_ptr(),
                             val.data_ptr(), mn.data_ptr(), mx.data_ptr(), _stream()))
    return (idx, val, mn, mx) if want_minmax else (idx, val)


def row_minmax(scores, n: Optional[int] = None):
    """Per-row (min, max) of a device fp32 matrix (hrag_row_minmax)."""
    torch = _torch()
    lib = _lib.load()
    s = scores.contiguous()
    b, ld = s.shape
    mn = torch.empty((b,), dtype=torch.float32, device=s.device)
    mx = torch.empty((b,), dtype=torch.float32, device=s.device)
    check(lib.hrag_row_minmax(s.data_ptr(), b, ld if n is None else n, ld, mn.data_ptr(), mx.data_ptr(),
                              _stream()))
    return mn, mx


class EngineStages:
    """The stage-level operators of include/hrag.h on caller-owned torch buffers -- what
    hipporag_amd.dist composes with its exchange steps.  One instance per (possibly row-sharded)
    engine."""

    def __init__(self, engine: HippoRAGEngine):
        self.e = engine
        self.lib = engine._lib
        self.h = engine._handle
        self.device = engine.device

    # --- metadata
    def layout(self, batch):
        return self.e.layout(batch)

    def new_state(self, batch):
        bc, ns = self.layout(batch)
        return _torch().zeros((ns, self.e.num_vertices, bc), dtype=_torch().float32, device=self.device)

    # --- similarity / selection
    def sim_scores(self, which, q):
        return self.e.sim_scores(which, q)

    def topk(self, scores, k, idx_offset=0):
        return topk_rows(scores, k, idx_offset=idx_offset, want_minmax=True)

    def row_minmax(self, scores):
        return row_minmax(scores)

    # --- reset vector
    def seeds(self, kept_idx, kept_score, kept_count, link_top_k):
        torch = _torch()
        b, kf = kept_idx.shape
        sv = torch.zeros((b, SEED_STRIDE), dtype=torch.int32, device=self.device)
        sw = torch.zeros((b, SEED_STRIDE), dtype=torch.float32, device=self.device)
        sc = torch.zeros((b,), dtype=torch.int32, device=self.device)
        flags = torch.zeros((b,), dtype=torch.int32, device=self.device)
        check(self.lib.hrag_stage_seeds(self.h, kept_idx.data_ptr(), kept_score.data_ptr(),
                                        kept_count.data_ptr(), kf, link_top_k, b, sv.data_ptr(),
                                        sw.data_ptr(), sc.data_ptr(), flags.data_ptr(), _stream()))
        return sv, sw, sc, flags

    def teleport(self, scores_full, mn, mx, weight, flags):
        torch = _torch()
        b, ld = scores_full.shape
        bc, ns = self.layout(b)
        tele = torch.empty((ns, self.e.n_passages, bc), dtype=torch.float32, device=self.device)
        check(self.lib.hrag_stage_teleport(self.h, scores_full.data_ptr(), ld, mn.data_ptr(), mx.data_ptr(),
                                           weight, flags.data_ptr(), b, tele.data_ptr(), _stream()))
        return tele

    # --- PPR on the owned rows
    def ppr_init(self, tele, seeds, batch, x):
        sv, sw, sc = seeds
        check(self.lib.hrag_stage_ppr_init(self.h, tele.data_ptr(), sv.data_ptr(), sw.data_ptr(),
                                           sc.data_ptr(), batch, x.data_ptr(), _stream()))

    def ppr_step(self, tele, seeds, batch, damping, x, y):
        sv, sw, sc = seeds
        check(self.lib.hrag_stage_ppr_step(self.h, tele.data_ptr(), sv.data_ptr(), sw.data_ptr(),
                                           sc.data_ptr(), batch, damping, x.data_ptr(), y.data_ptr(),
                                           _stream()))

    def colsum(self, x, batch):
        torch = _torch()
        nbytes = self.lib.hrag_colsum_workspace_bytes(self.h, batch)
        ws = torch.empty((max(nbytes, 8) // 8,), dtype=torch.float64, device=self.device)
        sums = torch.empty((batch,), dtype=torch.float64, device=self.device)
        check(self.lib.hrag_stage_colsum(self.h, x.data_ptr(), batch, ws.data_ptr(), sums.data_ptr(), _stream()))
        return sums

    def doc_scores(self, x, sums, batch, scores_full, mn, mx, flags):
        torch = _torch()
        out = torch.empty((batch, self.e.n_passages), dtype=torch.float32, device=self.device)
        check(self.lib.hrag_stage_doc_scores(self.h, x.data_ptr(), sums.data_ptr(), batch,
                                             scores_full.data_ptr(), scores_full.shape[1], mn.data_ptr(),
                                             mx.data_ptr(), flags.data_ptr(), out.data_ptr(),
                                             self.e.n_passages, _stream()))
        return out


def fp8_stage_plan(iters: int, damping: float = 0.5, tol: float = 0.0):
    """Python mirror of ppr8_plan (csrc/shard.hip): the stage lengths of the plain staged-fp8 PPR for `iters` sweeps.
    bench.py prices its per-instantiation launch times with it and tools/exp_fp8_final.py emulates it; the library
    never calls this.  iters < 19 or damping < 0.46: 1, 2, 3-sweep stages, remainder last.  Otherwise 1, 2, 3, 3-sweep
    stages, as many 4-sweep stages as fit, then: a 2-sweep stage last at a fixed count (20 = 1+2+3+4+4+4+2: five
    boundaries instead of six); 2 + 1 under a tolerance (20 = 1+2+3+3+4+4+2+1: the final measure reads lower)."""
    if iters < 19 or not damping >= 0.46:
        left = iters - 3
        return [1, 2] + [3] * (left // 3) + ([left % 3] if left % 3 else [])
    t = iters - (9 if tol > 0 else 8)
    a = t // 4
    while a > 0 and (t - 4 * a) % 3:
        a -= 1
    return [1, 2, 3] + [3] * ((t - 4 * a) // 3) + [4] * a + ([2, 1] if tol > 0 else [2])


class ShardStages(EngineStages):
    """The hrag_shard_* operators of a row-shard engine (staged fp8 PPR state) -- what
    hipporag_amd.dist.ShardedRetriever composes with its exchange steps."""

    def shard_layout(self, batch, groups=0):
        return self.e.shard_layout(batch, groups)

    def new_state(self, lay):           # one e4m3 state buffer, zero (row V of every group must stay zero)
        return _torch().zeros((lay.state_bytes,), dtype=_torch().uint8, device=self.device)

    def shard_score_facts(self, q_fact, k):
        return self.e.shard_score_facts(q_fact, k)

    def shard_passage_scores(self, q_pass):
        return self.e.shard_passage_scores(q_pass)

    def shard_prior_stats(self, mn, mx, weight, flags):
        return self.e.shard_prior_stats(mn, mx, weight, flags)

    def shard_ppr_begin(self, *a):
        return self.e.shard_ppr_begin(*a)

    def shard_ppr_sweep(self, sweep, group):
        return self.e.shard_ppr_sweep(sweep, group)

    def shard_ppr_step(self, step, group):
        return self.e.shard_ppr_step(step, group)

    def shard_ppr_est(self, est=None):
        return self.e.shard_ppr_est(est)

    def shard_ppr_decide(self, step):
        return self.e.shard_ppr_decide(step)

    def shard_ppr_gate_open(self, step):
        return self.e.shard_ppr_gate_open(step)

    def shard_finish(self, mn, mx, flags, k, want_residual=False):
        return self.e.shard_finish(mn, mx, flags, k, want_residual)
