#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_reference_order.py -x -q -k "wave_shape or whole_solves or live or random" 2>&1 | tail -5 > $O/r04_t2.txt
{ ORDER=ref timeout 300 python scripts/profile_phases.py 3 4096; } > $O/r04_wave_time2.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/ref_order_time.py 3 4096"
pass() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/r04b_pmc_ref_${name} -- $CMD > $O/r04b_pmc_ref_${name}.log 2>&1; echo "pmc $name rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
pass sq2 SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
pass sq3 SQ_ACTIVE_INST_LDS SQ_LDS_IDLE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES
