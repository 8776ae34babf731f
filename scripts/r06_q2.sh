mkdir -p gpurun_out
(
ORDER=ref DFTPAV_REF_SHAPE=quad timeout 300 python scripts/profile_phases.py 3 8
ORDER=ref DFTPAV_REF_SHAPE=quad timeout 300 python scripts/profile_phases.py 3 8192
for shape in wave quad; do for B in 8192 16384; do
DFTPAV_REF_SHAPE=$shape timeout 300 python scripts/ref_order_time.py 3 $B 2>&1 | grep "reference order"
done; done
) > gpurun_out/q2.log 2>&1
tail -70 gpurun_out/q2.log
