#!/bin/bash
# GPU box: exp / log of the reference order's moving-obstacle term with a quick phase in front (cr_trig.h) -- time and bits
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 200 python scripts/ref_order_batches.py cfg5_b1024 > $O/r05_crq_batches.txt 2>&1; cat $O/r05_crq_batches.txt
timeout 600 python -m pytest tests/test_gpu_reference_order.py -x -q 2>&1 | tail -4 > $O/r05_crq_tests.txt; echo "tests at $SECONDS s"; cat $O/r05_crq_tests.txt
