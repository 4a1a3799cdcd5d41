mkdir -p gpurun_out/r04i
for args in "1048576 20 0 8 1 32768" "1048576 20 0 8 0 32768" "1048576 20 64 16 1 32768" "1048576 20 0 4 1 32768"; do
timeout 300 tools/_bin/spmv_lds_bench $args >> gpurun_out/r04i/spmv_lds2.txt 2>&1
done
cat gpurun_out/r04i/spmv_lds2.txt
