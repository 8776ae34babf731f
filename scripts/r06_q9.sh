mkdir -p gpurun_out
(
timeout 1500 python -m pytest tests/test_gpu_reference_order.py -x -q -m gpu -k "moving or configs4 or gear_shifts or baseline_batch or obstacles or correctly_rounded" 2>&1 | tail -4
timeout 600 python scripts/ref_order_time.py 5 1024 2>&1 | grep "reference order"
ORDER=ref timeout 900 python scripts/profile_phases.py 5 1024 | grep -v "^$"
) > gpurun_out/q9.log 2>&1
cat gpurun_out/q9.log
