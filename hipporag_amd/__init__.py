"""hipporag_amd -- MI355X (gfx950) implementation of HippoRAG's retrieval hot path.

    from hipporag_amd import HippoRAG, QuerySolution, Chunk, RetrievalResult    # reference surface
    from hipporag_amd import HippoRAGEngine                                     # array-level engine

The compute lives in libhrag.so (hand-written HIP, C ABI in include/hrag.h); this package is the
host side: ctypes binding, torch plumbing (device memory, streams, torch.distributed) and the mirror
of the reference's Python surface for this path.  Nothing here falls back to the CPU.
"""

from .retriever import (Chunk, HippoRAG, QuerySolution, RetrievalConfig, RetrievalResult,  # noqa: F401
                        compute_mdhash_id, min_max_normalize, text_processing)
from .graph import CSRGraph, build_csr, bf16_bits_to_float, float_to_bf16_bits  # noqa: F401


def __getattr__(name):
    # the engine imports torch; keep `import hipporag_amd` light
    if name in ("HippoRAGEngine", "EngineStages", "topk_rows", "row_minmax"):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)
