mkdir -p gpurun_out
(
DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_order_time.py 3 2048 4096 8192 2>&1 | grep "reference order"
DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 2 4 8 2>&1 | grep shape
for slice in 32 128; do echo slice $slice; DFTPAV_REF_SLICE=$slice DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape; done
for slots in 384 640; do echo slots $slots; DFTPAV_REF_SLOTS=$slots DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape; done
) > gpurun_out/q5.log 2>&1
cat gpurun_out/q5.log
