#!/bin/bash
# One GPU's share of BASELINE configs[4] through bench.py INCLUDING the CPU oracle spot check (needs ~100 GB
# of host memory for the fp32 copies of the embeddings and the scipy graph): runs only if the box has it.
mkdir -p gpurun_out/cfg5
avail_kb=$(awk '/MemAvailable/ {print $2}' /proc/meminfo)
lim=$(cat /sys/fs/cgroup/memory.max 2>/dev/null || cat /sys/fs/cgroup/memory/memory.limit_in_bytes 2>/dev/null || echo max)
echo "MemAvailable ${avail_kb} kB, cgroup limit ${lim}" | tee gpurun_out/cfg5/mem.log
if [ "$avail_kb" -lt 300000000 ]; then echo "not enough host memory: skipping the oracle leg"; exit 0; fi
if [ "$lim" != "max" ] && [ "$lim" -lt 300000000000 ]; then echo "cgroup limit too low: skipping"; exit 0; fi
(time timeout 1000 python bench.py --config cfg5gpu --steps 3 --warmup 1 --cpu-queries 2 --cpu-budget-s 60 --sweep-launches 5) > gpurun_out/cfg5/bench_oracle.log 2>&1
tail -c 1500 gpurun_out/cfg5/bench_oracle.log
