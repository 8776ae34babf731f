mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_reference_order.py -x -q -s --durations=5 > gpurun_out/r3_reftests18.txt 2>&1
tail -30 gpurun_out/r3_reftests18.txt
timeout 600 python scripts/ref_order_time.py 3 32 256 4096 > gpurun_out/r3_reftime18.txt 2>&1; cat gpurun_out/r3_reftime18.txt
