#!/usr/bin/env python
"""Write the column indices of a benchmark graph's CSR (row-major, int32) for `membench --cols`:
    python tools/dump_cols.py --config cfg3 --out /tmp/cols.bin"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from bench import CONFIGS
from hipporag_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg3")
ap.add_argument("--out", default="/tmp/cols.bin")
args = ap.parse_args()
cfg = CONFIGS[args.config]
kg = synth.make_kg(cfg["V"], cfg["E"], cfg["seed"], power_law=bool(cfg.get("power_law")))
np.ascontiguousarray(kg.csr.col_idx, dtype=np.int32).tofile(args.out)
print(kg.num_vertices, kg.csr.nnz)
