mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_order.py -x -q > gpurun_out/r3_reftests12.txt 2>&1
tail -3 gpurun_out/r3_reftests12.txt
timeout 600 python scripts/ref_order_time.py 3 32 256 2048 4096 > gpurun_out/r3_reftime12.txt 2>&1
cat gpurun_out/r3_reftime12.txt
