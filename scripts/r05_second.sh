#!/bin/bash
# GPU box, round 5: the reference order at the BASELINE batch sizes (default library and the 40-term kernel built for 256
# registers), then the GPU tests added this round
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 300 python scripts/ref_order_batches.py > $O/r05_ref_batches.txt 2>&1; echo "batches rc=$? at $SECONDS s"
DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_nc40.so timeout 200 python scripts/ref_order_batches.py cfg2_b4096 >> $O/r05_ref_batches.txt 2>&1; echo "variant rc=$? at $SECONDS s"
cat $O/r05_ref_batches.txt
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_reference_order.py tests/test_gpu_parity.py -q -x -k "restarts_behind or baseline_batch_sizes or reference_builds_vectors" 2>&1 | tail -12 > $O/r05_new_tests.txt; echo "tests rc=$? at $SECONDS s"; cat $O/r05_new_tests.txt
