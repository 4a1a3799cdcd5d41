set -x
mkdir -p gpurun_out/r04a
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size" -s 2>&1 | tail -15 > gpurun_out/r04a/tests_fullsize.log
python -m pytest tests/test_gpu_fp8_adversarial.py tests/test_mirror_surface.py tests/test_gpu_shard.py -m gpu -x -q -s 2>&1 | tail -30 > gpurun_out/r04a/tests_adv.log
python bench.py --steps 10 --warmup 2 --cpu-queries 6 --cpu-budget-s 8 --cpu-vec-queries 8 > gpurun_out/r04a/bench_cfg3.json 2> gpurun_out/r04a/bench_cfg3.err
HRAG_FORCE_DIST=1 python bench.py --config cfg4 --steps 2 --warmup 1 > gpurun_out/r04a/bench_cfg4_world1.json 2> gpurun_out/r04a/bench_cfg4_world1.err
python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r04a/bench_gpus2.out 2> gpurun_out/r04a/bench_gpus2.err; echo "rc=$?" >> gpurun_out/r04a/bench_gpus2.err
tail -3 gpurun_out/r04a/*.log; tail -c 600 gpurun_out/r04a/bench_cfg3.json; tail -5 gpurun_out/r04a/bench_gpus2.err
