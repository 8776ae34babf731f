#!/bin/bash
# GPU box: the value line at pipeline depths 2 (default), 3, 4, 6
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for d in 2 3 4 6 8; do
  DFTPAV_BENCH_DEPTH=$d timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('depth $d: value %.0f ms/step %.1f frac %.3f steps %d to_result %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['steps'], d.get('time_to_result_ms')))
"
done > $O/r05_depth.txt 2>&1
cat $O/r05_depth.txt
