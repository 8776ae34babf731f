#!/bin/bash
# GPU box, the first call of round 5: what round 4 wrote or changed after its last full GPU run and could not run itself.
#   1. the dense direction (csrc/dense_dir.h): parity with oracle order 3 + the stored vectors, then the timing against the
#      two-loop recursion (scripts/dense_check.py);
#   2. the tests added at the end of round 4: the reference order against the reference's objects on the correctly rounded libm,
#      the drop-in's live case against that build, the reference-counted communicator;
#   3. the whole GPU suite.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 400 python scripts/dense_check.py parity > $O/r05_dense_parity.txt 2>&1; echo "dense parity rc=$? at $SECONDS s"; tail -3 $O/r05_dense_parity.txt
timeout 300 python scripts/dense_check.py time > $O/r05_dense_time.txt 2>&1; echo "dense time rc=$? at $SECONDS s"; cat $O/r05_dense_time.txt
DFTPAV_TEST_DENSE=1 timeout 600 python -m pytest tests/test_gpu_dense.py -q 2>&1 | tail -5 > $O/r05_dense_tests.txt; cat $O/r05_dense_tests.txt
timeout 600 python -m pytest tests/test_gpu_reference_order.py tests/test_gpu_dropin.py tests/test_gpu_dist.py -q -k "correctly_rounded or live_case or one_communicator" 2>&1 | tail -5 > $O/r05_new_tests.txt; echo "new tests rc=$? at $SECONDS s"; cat $O/r05_new_tests.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/r05_gpu_suite.txt; echo "suite rc=$? at $SECONDS s"; cat $O/r05_gpu_suite.txt
