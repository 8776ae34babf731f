#!/bin/bash
# GPU box: the reference order's column sweeps (band_cols) on top of the DPP chains -- bits and time
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 200 python scripts/ref_order_time.py 3 1 256 4096 > $O/r05_cols_time.txt 2>&1; timeout 100 python scripts/ref_order_time.py 2 1 >> $O/r05_cols_time.txt 2>&1; cat $O/r05_cols_time.txt
ORDER=ref timeout 200 python scripts/profile_phases.py 3 4096 > $O/r05_cols_phases.txt 2>&1; tail -14 $O/r05_cols_phases.txt
timeout 300 python scripts/ref_order_batches.py > $O/r05_cols_batches.txt 2>&1; cat $O/r05_cols_batches.txt
timeout 900 python -m pytest tests/test_gpu_reference_order.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -6 > $O/r05_cols_tests.txt; echo "tests rc=$? at $SECONDS s"; cat $O/r05_cols_tests.txt
