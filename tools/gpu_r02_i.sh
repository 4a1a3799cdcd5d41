#!/bin/bash
# round 2: kernel timelines of one retrieve at cfg2 (B = 64) and cfg3 B = 1 (where do the microseconds go)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02i}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t_cfg2" -o t -- python "$REPO/tools/trace_target.py" --config cfg2 > "$OUT/t_cfg2.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t_b1" -o t -- python "$REPO/tools/trace_target.py" --config cfg3 --batch 1 > "$OUT/t_b1.log" 2>&1
cd "$REPO"
for T in t_cfg2 t_b1; do
  F=$(find "$OUT/$T" -name '*kernel_trace.csv' | head -1)
  python tools/timeline.py "$F" 4 > "$OUT/$T.timeline.txt" 2>&1
  tail -3 "$OUT/$T.timeline.txt"
  find "$OUT/$T" -name '*.csv' -size +2M -delete
done
