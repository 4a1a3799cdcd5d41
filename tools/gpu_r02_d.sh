#!/bin/bash
# round 2: row-order experiment -- mode-C launch time + TCC hit / miss + FETCH_SIZE under the three SELL-8 row orders
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r02d}
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python tools/exp_row_order.py > "$OUT/row_order_timing.json" 2> "$OUT/row_order_timing.err"
tail -c 600 "$OUT/row_order_timing.json"; tail -2 "$OUT/row_order_timing.err"
cd /tmp && export TMPDIR=/tmp
for F in 0 64 128; do
  for C in "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE; do
    N=$(echo $C | tr ' ' '_')
    HRAG_FLAGS=$F timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/order$F/pmc_$N" -o pmc -- \
        python "$REPO/tools/pmc_target.py" > "$OUT/order${F}_$N.log" 2>&1
  done
done
cd "$REPO"; find "$OUT" -name '*counter_collection.csv' | head
