#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_psw.so ORDER=ref timeout 300 python scripts/profile_phases.py 3 4096 > $O/r04_psw.txt 2>&1
DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_psw.so ORDER=ref timeout 300 python scripts/profile_phases.py 3 64 >> $O/r04_psw.txt 2>&1
