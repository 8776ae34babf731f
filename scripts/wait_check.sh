#!/bin/bash
# Developer script: compiles <dir>/solver.hip (with its headers beside it) to assembly and reports, for every cluster of >= 12 global loads in the
# static LV=5 solve kernel, how far behind it the next s_waitcnt vmcnt sits and what it waits for (a wait right behind a cluster = no overlap of the
# memory latency with the arithmetic that follows).   scripts/wait_check.sh <dir>
d=$1
cd $d && /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-sched-strategy=max-ilp --cuda-device-only -S solver.hip -o k.s 2>&1 | grep error
awk '/^_ZN6dftpav13solver_kernelILb0ELi5ELi512EEEvPKNS_8DevBatchEiNS_9SchedArgsE:/,/s_endpgm/' k.s | grep -v "^\s*;" | grep -v "^\s*\." > k0.s
python3 - <<'PY'
import re
L=[l.strip() for l in open('k0.s')]
# find runs: >=12 global_load_dwordx2 within 80 lines
idx=[i for i,l in enumerate(L) if l.startswith('global_load_dwordx2')]
i=0; out=[]
while i<len(idx):
    j=i
    while j+1<len(idx) and idx[j+1]-idx[j]<30: j+=1
    if j-i+1>=12:
        last=idx[j]
        for k in range(last+1,len(L)):
            m=re.match(r's_waitcnt vmcnt\((\d+)\)',L[k])
            if m: out.append((idx[i],j-i+1,k-last,int(m.group(1)))); break
    i=j+1
print("clusters (line, loads, distance to next vmcnt wait, N):",out)
PY
grep -A12 "^\s*\.name:\s*_ZN6dftpav13solver_kernelILb0ELi5" k.s | grep -E "vgpr_spill|private_segment_fixed" | tr '\n' ' '; echo
