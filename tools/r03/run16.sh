#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/r03/dbg_knn.py 2>&1 | tail -5
