#!/bin/bash
# GPU box: A/B of two builds of the library on configs[4] (dftpav_amd/libdftpav_hip_before.so = the build before the change)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export GPU_MAX_HW_QUEUES=16
for lib in libdftpav_hip_before.so libdftpav_hip.so libdftpav_hip_before.so libdftpav_hip.so; do DFTPAV_LIB=$R/dftpav_amd/$lib timeout 60 python scripts/cfg5_streams.py 2>&1 | tail -1; done > gpurun_out/r04_cfg5_ab.txt
cat gpurun_out/r04_cfg5_ab.txt
