#!/bin/bash
# GPU box, round 5: the whole GPU test suite and smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/r05_gpu_suite.txt; echo "suite rc=$? at $SECONDS s"; cat $O/r05_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.txt 2>&1; echo "smoke rc=$? at $SECONDS s"; tail -3 $O/r05_smoke.txt
