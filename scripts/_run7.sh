mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reference_order.py -x -q --durations=5 > gpurun_out/r3_reftests7.txt 2>&1
tail -25 gpurun_out/r3_reftests7.txt
