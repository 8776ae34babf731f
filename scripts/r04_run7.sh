#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
rocprofv3 --list-avail > $O/r04_counters.txt 2>&1
ls /opt/rocm/lib | grep -i "trace-decoder\|att" > $O/r04_att_libs.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/r04_t8.txt
