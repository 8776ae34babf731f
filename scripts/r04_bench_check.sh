#!/bin/bash
# GPU box: the default bench line of the tree (after the split of bench.py:main), timed
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
SECONDS=0
timeout 300 python bench.py > $O/bench_line_r04_split.json 2> $O/bench_line_r04_split.err; echo "bench rc=$? after $SECONDS s"
tail -c 600 $O/bench_line_r04_split.err
