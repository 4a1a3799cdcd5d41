mkdir -p gpurun_out/r04j
python -m pytest tests/test_gpu_accel.py tests/test_gpu_fp8_adversarial.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r04j/tests.log
python bench.py --steps 10 --warmup 2 --cpu-queries 8 --cpu-budget-s 8 --cpu-vec-queries 8 > gpurun_out/r04j/bench_cfg3.json 2> gpurun_out/r04j/bench_cfg3.err
cat gpurun_out/r04j/tests.log
