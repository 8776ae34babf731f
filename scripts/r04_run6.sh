#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -8 > $O/r04_t7.txt
{ ORDER=ref timeout 300 python scripts/profile_phases.py 3 4096 2>&1 | grep -v "^x (exp\|^init\|^misc"; } > $O/r04_wave_time7.txt 2>&1
timeout 900 python bench.py --steps 8 --warmup 2 > $O/r04_bench1.json 2> $O/r04_bench1.err
