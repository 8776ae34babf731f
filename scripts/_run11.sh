mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_reference_order.py -x -q -s --durations=5 > gpurun_out/r3_reftests11.txt 2>&1
tail -30 gpurun_out/r3_reftests11.txt
