#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
{
for sl in 32 64 128 256 512; do echo "=== slice $sl"; DFTPAV_REF_SLICE=$sl timeout 200 python scripts/ref_order_time.py 3 4096 2>&1 | grep "reference order"; done
for w in 4 2; do echo "=== waves per workgroup $w"; DFTPAV_REF_WAVES=$w timeout 200 python scripts/ref_order_time.py 3 4096 2>&1 | grep "reference order"; done
echo "=== TEAM shape at 1024 / WAVE shape at 1024"; DFTPAV_REF_SHAPE=team timeout 200 python scripts/ref_order_time.py 3 1024 2>&1 | grep "reference order"; DFTPAV_REF_SHAPE=wave timeout 200 python scripts/ref_order_time.py 3 1024 2>&1 | grep "reference order"
echo "=== TEAM / WAVE at 512"; DFTPAV_REF_SHAPE=team timeout 200 python scripts/ref_order_time.py 3 512 2>&1 | grep "reference order"; DFTPAV_REF_SHAPE=wave timeout 200 python scripts/ref_order_time.py 3 512 2>&1 | grep "reference order"
echo "=== TEAM / WAVE at 2048"; DFTPAV_REF_SHAPE=team timeout 200 python scripts/ref_order_time.py 3 2048 2>&1 | grep "reference order"; DFTPAV_REF_SHAPE=wave timeout 200 python scripts/ref_order_time.py 3 2048 2>&1 | grep "reference order"
} > $O/r04_scan.txt 2>&1
