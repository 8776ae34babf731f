#!/bin/bash
# GPU box: configs[4] after the pair chain pass was restricted to the pieces of its batch: kernel time + phases, then the
# parity tests that run the moving-obstacle kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 120 python scripts/cfg5_time.py > $O/r04_cfg5_chain.txt 2>&1; echo "cfg5 rc=$? at $SECONDS s"
timeout 110 python -m pytest tests/test_gpu_parity.py -x -q -k "configs4 or moving_obstacles_that or 5-2 or 5-6 or fit_surround" 2>&1 | tail -4 > $O/r04_cfg5_chain_tests.txt; echo "tests rc=$? at $SECONDS s"
cat $O/r04_cfg5_chain_tests.txt; head -3 $O/r04_cfg5_chain.txt | cut -c1-300
