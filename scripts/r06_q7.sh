mkdir -p gpurun_out
(
timeout 600 python -m pytest tests/test_gpu_reference_order.py -x -q -m gpu -k "quad" 2>&1 | tail -4
for hand in 0 256 768 1536; do echo "hand-over at $hand"; DFTPAV_REF_QUAD_HANDOVER=$hand DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_order_time.py 3 4096 2>&1 | grep "reference order"; done
STEPS=20 WARMUP=5 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape
STEPS=20 WARMUP=5 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape
) > gpurun_out/q7.log 2>&1
cat gpurun_out/q7.log
