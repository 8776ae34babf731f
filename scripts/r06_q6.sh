mkdir -p gpurun_out
(
timeout 300 python scripts/quad_check.py 64 | grep -i "mismatch\|failed\|OK"
ORDER=ref DFTPAV_REF_SHAPE=quad timeout 300 python scripts/profile_phases.py 3 8 | grep -v "^$"
STEPS=20 WARMUP=5 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape
STEPS=20 WARMUP=5 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape
STEPS=24 WARMUP=8 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 8 2>&1 | grep shape
) > gpurun_out/q6.log 2>&1
cat gpurun_out/q6.log
