#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/r04_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_final_smoke.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_r04_final.json 2> $O/bench_line_r04_final.err; echo "bench rc=$?"
