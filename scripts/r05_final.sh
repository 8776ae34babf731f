#!/bin/bash
# GPU box, round 5, final tree: the whole GPU suite and smoke(), then everything profiles/r05_* is summarised from
# (scripts/collect_profiles.sh: rocprofv3 kernel statistics and counter passes of a short run of bench.py, the phase timers, the
# default bench line), then a short fuzz of the reference order
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/r05_gpu_suite.txt; echo "suite rc=$? at $SECONDS s"; cat $O/r05_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.txt 2>&1; echo "smoke rc=$? at $SECONDS s"; tail -2 $O/r05_smoke.txt
bash scripts/collect_profiles.sh r05; echo "profiles done at $SECONDS s"
cd $R
timeout 400 python scripts/fuzz_reference_order.py 250 > $O/r05_fuzz_reference_order.txt 2>&1; echo "fuzz rc=$? at $SECONDS s"; tail -2 $O/r05_fuzz_reference_order.txt
head -c 600 $O/bench_line_r05.json; echo
