#!/bin/bash
# GPU box: A/B of two builds of the library on configs[4] (dftpav_amd/libdftpav_hip_before.so = the build before the change):
# the parity tests that run the moving-obstacle kernel, isolated batches alternately, then the 8-stream form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export GPU_MAX_HW_QUEUES=16
timeout 40 python -m pytest tests/test_gpu_parity.py -x -q -k "configs4 or moving_obstacles_that or 5-2 or 5-6 or fit_surround" 2>&1 | tail -2 > gpurun_out/r04_cfg5_ab.txt
for lib in libdftpav_hip_before.so libdftpav_hip.so libdftpav_hip_before.so libdftpav_hip.so; do DFTPAV_LIB=$R/dftpav_amd/$lib timeout 20 python scripts/cfg5_streams.py 1024 isolated 2>&1 | tail -1; done >> gpurun_out/r04_cfg5_ab.txt
for lib in libdftpav_hip_before.so libdftpav_hip.so; do DFTPAV_LIB=$R/dftpav_amd/$lib timeout 25 python scripts/cfg5_streams.py 1024 2>&1 | tail -1; done >> gpurun_out/r04_cfg5_ab.txt
cat gpurun_out/r04_cfg5_ab.txt
