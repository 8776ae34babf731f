mkdir -p gpurun_out
(
timeout 900 python -m pytest tests/test_gpu_reference_order.py -x -q -m gpu -k "quad or true_divisions" 2>&1 | tail -4
ORDER=ref DFTPAV_REF_SHAPE=quad timeout 300 python scripts/profile_phases.py 3 8 | grep "two-loop\|samples\|kernel ms"
STEPS=20 WARMUP=5 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape
STEPS=20 WARMUP=5 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape
DFTPAV_REF_EXACT_DIV=1 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/quad_check.py 64 | grep -i "mismatch\|failed\|OK"
) > gpurun_out/q8.log 2>&1
cat gpurun_out/q8.log
