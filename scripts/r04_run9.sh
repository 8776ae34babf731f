#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
{
echo "=== out of line (default)"; timeout 300 python scripts/cfg5_time.py 1024 2>&1 | grep -v "^$"
echo "=== inline + rejection"; DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_pinl.so timeout 300 python scripts/cfg5_time.py 1024 2>&1 | grep -v "^$"
for ho in 384 512; do echo "=== inline, hand-over $ho"; DFTPAV_HANDOVER=$ho DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_pinl.so timeout 300 python scripts/cfg5_time.py 1024 2>&1 | head -3; done
for sl in 16 64; do echo "=== inline, slice $sl"; DFTPAV_SLICE=$sl DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_pinl.so timeout 300 python scripts/cfg5_time.py 1024 2>&1 | head -3; done
echo "=== inline, 128 threads"; DFTPAV_THREADS=128 DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_pinl.so timeout 300 python scripts/cfg5_time.py 1024 2>&1 | head -3
} > $O/r04_cfg5_variants.txt 2>&1
