#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
{
for v in "" tlinl two evno ilp twoinl o2; do
  if [ -n "$v" ]; then export DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_$v.so; else unset DFTPAV_LIB; fi
  echo "=== variant '$v'"
  ORDER=ref timeout 300 python scripts/profile_phases.py 3 4096 2>&1 | grep -v "^x (exp\|^init\|^misc"
done
} > $O/r04_variants.txt 2>&1
unset DFTPAV_LIB
timeout 900 python -m pytest tests/test_gpu_reference_order.py -x -q -k "wave_shape or whole_solves or live" 2>&1 | tail -5 > $O/r04_t4.txt
