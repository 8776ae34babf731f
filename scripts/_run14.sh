mkdir -p gpurun_out
for t in 256 128; do
  echo "== threads $t"
  DFTPAV_REF_THREADS=$t timeout 600 python -m pytest tests/test_gpu_reference_order.py -x -q 2>&1 | tail -2
  DFTPAV_REF_THREADS=$t timeout 600 python scripts/ref_order_time.py 3 32 256 2048 4096 2>&1 | grep "reference order"
done > gpurun_out/r3_reftime14.txt 2>&1
cat gpurun_out/r3_reftime14.txt
