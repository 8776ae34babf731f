#!/bin/bash
# GPU box, round 5: everything profiles/r05_* is summarised from (scripts/collect_profiles.sh: rocprofv3 kernel statistics and
# counter passes of a short run of bench.py, the phase timers, the default bench line), then a short fuzz of the reference order
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
bash scripts/collect_profiles.sh r05; echo "profiles done at $SECONDS s"
cd $R
timeout 400 python scripts/fuzz_reference_order.py 250 > $O/r05_fuzz_reference_order.txt 2>&1; echo "fuzz rc=$? at $SECONDS s"; tail -3 $O/r05_fuzz_reference_order.txt
head -c 1500 $O/bench_line_r05.json; echo; tail -5 $O/bench_line_r05.err
