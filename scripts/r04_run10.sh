#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_reference_order.py -x -q 2>&1 | tail -5 > $O/r04_t10.txt
{
for v in "" tlmask; do
  if [ -n "$v" ]; then export DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_$v.so; else unset DFTPAV_LIB; fi
  echo "=== variant '$v'"
  ORDER=ref timeout 300 python scripts/profile_phases.py 3 4096 2>&1 | grep -v "^x (exp\|^init\|^misc"
done
} > $O/r04_variants2.txt 2>&1
