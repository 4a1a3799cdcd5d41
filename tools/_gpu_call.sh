set -x
mkdir -p gpurun_out/r04g
python -m pytest tests/test_gpu_accel.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r04g/tests_accel.log
tail -n 30 gpurun_out/r04g/tests_accel.log
