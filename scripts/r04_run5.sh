#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_reference_order.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -8 > $O/r04_t6.txt
{ ORDER=ref timeout 300 python scripts/profile_phases.py 3 4096 2>&1 | grep -v "^x (exp\|^init\|^misc"; } > $O/r04_wave_time6.txt 2>&1
