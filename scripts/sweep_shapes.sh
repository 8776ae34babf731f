export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.0f ms/step %.1f frac %.3f kernel_ms %.1f p50 %.1f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['p50_ms_per_solve']))
    elif 'rror' in l: print(l.strip())
"; }
run A=0
run DFTPAV_THREADS=64 DFTPAV_PPT=1 DFTPAV_SLOTS=1536
run DFTPAV_THREADS=64 DFTPAV_PPT=2 DFTPAV_SLOTS=1280
run DFTPAV_THREADS=64 DFTPAV_PPT=1 DFTPAV_SLOTS=1536 DFTPAV_SLICE=96
run DFTPAV_THREADS=64 DFTPAV_PPT=1 DFTPAV_SLOTS=2048
run DFTPAV_THREADS=128 DFTPAV_SLOTS=1280
