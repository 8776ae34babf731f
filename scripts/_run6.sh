mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_order.py -x -q > gpurun_out/r3_reftests3.txt 2>&1
timeout 600 python scripts/ref_order_time.py 3 1 32 256 2048 > gpurun_out/r3_reftime.txt 2>&1
tail -3 gpurun_out/r3_reftests3.txt; cat gpurun_out/r3_reftime.txt
