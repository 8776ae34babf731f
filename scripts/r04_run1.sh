#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_reference_order.py -x -q 2>&1 | tail -15 > $O/r04_t1.txt
{ DFTPAV_VERBOSE=1 timeout 300 python scripts/ref_order_time.py 3 4096 2>&1 | grep -v "launch mode" | head -20; ORDER=ref timeout 300 python scripts/profile_phases.py 3 4096; } > $O/r04_wave_time.txt 2>&1
