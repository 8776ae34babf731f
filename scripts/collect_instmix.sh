#!/bin/bash
# GPU box: dynamic instruction mix by class of the two solve kernels from the SQ_INSTS_* counters (rocprofv3 --pmc passes of their
# own; neither PC sampling nor the thread-trace decoder is available for gfx950 in this image -- `rocprofv3 --pc-sampling-*`
# answers "configuration not supported on any of the agents", /opt/rocm/lib has no trace decoder).
#   scripts/collect_instmix.sh r04     then, here:  python scripts/summarize_instmix.py r04
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
DEV="python $R/bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0"
REF="python $R/scripts/ref_order_time.py 3 4096"
pass() { k=$1; name=$2; shift 2; cmd=$DEV; [ $k = ref ] && cmd=$REF; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $O/mix_${k}_${name}_$tag -- $cmd > $O/mix_${k}_${name}_$tag.log 2>&1; echo "mix $k $name rc=$?"; }
for k in dev ref; do
  pass $k a SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
  pass $k b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH
  pass $k c SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32
done
pass ref fetch FETCH_SIZE
pass ref write WRITE_SIZE
pass ref sq2 SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refwave_$tag -- $REF > $O/prof_refwave_$tag.log 2>&1; echo "ref kernel trace rc=$?"
cd $R
timeout 600 python -m pytest tests/test_gpu_lockstep.py -s -q 2>&1 | grep -v "^$" > $O/lockstep_$tag.txt
