#!/usr/bin/env python
"""Run on a machine WITH python-igraph (the reference's environment): dump the edge list of a HippoRAG
``graph.pickle`` (reference src/hipporag/HippoRAG.py:1225-1230) to the ``.npz`` that
``hipporag_amd.loaders.load_reference_workdir(graph_edges=...)`` reads.

    python tools/export_igraph_edges.py outputs/<llm>_<emb>/graph.pickle graph_edges.npz
"""
import sys

import numpy as np


def main():
    import igraph as ig                         # not available in the build container
    g = ig.Graph.Read_Pickle(sys.argv[1])
    es = np.asarray(g.get_edgelist(), dtype=np.int64).reshape(-1, 2)
    np.savez_compressed(sys.argv[2], names=np.asarray(g.vs["name"], dtype=str), src=es[:, 0], dst=es[:, 1],
                        weight=np.asarray(g.es["weight"], dtype=np.float64))
    print(f"{g.vcount()} vertices, {g.ecount()} edges -> {sys.argv[2]}")


if __name__ == "__main__":
    main()
