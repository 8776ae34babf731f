mkdir -p gpurun_out
(
for slots in 256 512 768; do for slice in 64 256; do
echo "slots $slots slice $slice"
DFTPAV_REF_SLOTS=$slots DFTPAV_REF_SLICE=$slice DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 4 2>&1 | grep shape
done; done
echo "slots 512 slice 64 depth 2, 8"
DFTPAV_REF_SLOTS=512 DFTPAV_REF_SLICE=64 DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_stream_time.py 2 8 2>&1 | grep shape
) > gpurun_out/q4.log 2>&1
cat gpurun_out/q4.log
