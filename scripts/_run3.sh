mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_order.py -x -q > gpurun_out/r3_reftests3.txt 2>&1
ORDER=ref timeout 600 python scripts/profile_phases.py 3 32 > gpurun_out/r3_refprof_b.txt 2>&1
ORDER=ref timeout 600 python scripts/profile_phases.py 1 32 >> gpurun_out/r3_refprof_b.txt 2>&1
tail -5 gpurun_out/r3_reftests3.txt; cat gpurun_out/r3_refprof_b.txt
