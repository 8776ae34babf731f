# Developer script (GPU box): the bench (8 steps) under overrides of the schedule.   bash scripts/sweep_shapes.sh
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --cpu-sample 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f ms/step %.1f frac %.3f kernel_ms %.1f p50 %.1f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['p50_ms_per_solve']))
    elif 'rror' in l: print(l.strip())
"; }
run A=0
run DFTPAV_SLICE=64
run DFTPAV_SLICE=96
run DFTPAV_SLICE=192
run DFTPAV_SLICE=256
run DFTPAV_SLOTS=1792
run DFTPAV_SLOTS=2304
run A=1
