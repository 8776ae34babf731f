mkdir -p gpurun_out
(
timeout 300 python scripts/quad_check.py 64 | grep -i "mismatch\|failed\|OK\|kernel ms"
ORDER=ref DFTPAV_REF_SHAPE=quad timeout 300 python scripts/profile_phases.py 3 8
ORDER=ref DFTPAV_REF_SHAPE=quad timeout 300 python scripts/profile_phases.py 3 4096
for w in 1 2 4; do for B in 4096 8192 16384; do
echo "waves per workgroup $w"; DFTPAV_REF_QUAD_WAVES=$w DFTPAV_REF_SHAPE=quad timeout 300 python scripts/ref_order_time.py 3 $B 2>&1 | grep "reference order"
done; done
) > gpurun_out/q3.log 2>&1
tail -80 gpurun_out/q3.log
