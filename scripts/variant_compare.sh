#!/bin/bash
# Developer script (GPU box): phase profile + short bench for the shipped library and each variant given.
export TMPDIR=/tmp
for v in "" "$@"; do
  if [ -n "$v" ]; then export DFTPAV_LIB=$PWD/dftpav_amd/variants/libdftpav_hip_$v.so; else unset DFTPAV_LIB; fi
  echo "=========== variant '${v:-shipped}'"
  python scripts/profile_phases.py 3 4096 2>&1 | grep -v "^$" | tail -14
  python scripts/profile_phases.py 3 256 2>&1 | grep -E "two-loop|solves/s|E3|E4"
  python bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('BENCH value %.0f ms/step %.1f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
"
done
