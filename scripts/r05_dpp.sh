#!/bin/bash
# GPU box: the DPP chain of the reference order's sequential sums at every kernel width -- bits (the reference-order and drop-in
# tests) and time
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 300 python scripts/ref_order_batches.py > $O/r05_dpp_batches.txt 2>&1; echo "batches rc=$? at $SECONDS s"; cat $O/r05_dpp_batches.txt
timeout 200 python scripts/ref_order_time.py 2 1 > $O/r05_dpp_time.txt 2>&1; timeout 200 python scripts/ref_order_time.py 3 1 4096 >> $O/r05_dpp_time.txt 2>&1; cat $O/r05_dpp_time.txt
timeout 900 python -m pytest tests/test_gpu_reference_order.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -6 > $O/r05_dpp_tests.txt; echo "tests rc=$? at $SECONDS s"; cat $O/r05_dpp_tests.txt
