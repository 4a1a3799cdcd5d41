set -x
mkdir -p gpurun_out/r04e
python -m pytest tests/test_gpu_accel.py tests/test_gpu_fp8_adversarial.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r04e/tests_accel.log
python bench.py --steps 10 --warmup 2 --cpu-queries 12 --cpu-budget-s 12 --cpu-vec-queries 8 > gpurun_out/r04e/bench_cfg3.json 2> gpurun_out/r04e/bench_cfg3.err
tail -n 25 gpurun_out/r04e/tests_accel.log; tail -n 3 gpurun_out/r04e/bench_cfg3.err
