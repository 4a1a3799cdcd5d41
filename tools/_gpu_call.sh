mkdir -p gpurun_out/r04k
python tools/bench_mirror.py --config cfg3 --queries 1024 > gpurun_out/r04k/mirror_cfg3.json 2> gpurun_out/r04k/mirror_cfg3.err
python tools/bench_mirror.py --config cfg2 --queries 1024 > gpurun_out/r04k/mirror_cfg2.json 2> gpurun_out/r04k/mirror_cfg2.err
timeout 900 python bench.py --config cfg4local --cpu-queries 16 > gpurun_out/r04k/bench_cfg4local.json 2> gpurun_out/r04k/bench_cfg4local.err
timeout 1200 python bench.py --config cfg5gpu --steps 3 --warmup 1 --cpu-queries 8 --cpu-budget-s 400 --sweep-launches 5 > gpurun_out/r04k/bench_cfg5gpu.json 2> gpurun_out/r04k/bench_cfg5gpu.err
tail -c 400 gpurun_out/r04k/mirror_cfg3.json gpurun_out/r04k/mirror_cfg2.json; tail -c 600 gpurun_out/r04k/bench_cfg4local.json; tail -c 800 gpurun_out/r04k/bench_cfg5gpu.json; tail -n 3 gpurun_out/r04k/*.err
