#!/bin/bash
# GPU box: everything profiles/ is summarised from, for one tag (default r03).  rocprofv3 runs from /tmp with TMPDIR=/tmp;
# the counter passes are separate runs with --pmc only (never combined with trace domains).
#   scripts/collect_profiles.sh r04      then, here:  python scripts/summarize_profiles.py r04 ; python scripts/summarize_instmix.py r04
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0"
O=$R/gpurun_out
echo "$CMD" > $O/profile_command_$tag.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- $CMD > $O/prof_$tag.log 2>&1; echo "kernel trace rc=$?"
pass() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${name}_$tag -- $CMD > $O/pmc_${name}_$tag.log 2>&1; echo "pmc $name rc=$?"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
pass sq2 SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
# the reference-order kernel (solver_ref.hip): rocprofv3 kernel statistics of isolated solves at batch 256 and 4096, its own phase timer
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ref_$tag -- python $R/scripts/ref_order_time.py 3 256 4096 > $O/prof_ref_$tag.log 2>&1; echo "reference-order kernel trace rc=$?"
cd $R
# reference order: the WAVE shape at 4096 (one wave per trajectory, ring), the TEAM shape at 32, kernel times over the batch sizes
{ ORDER=ref timeout 600 python scripts/profile_phases.py 3 4096; ORDER=ref timeout 600 python scripts/profile_phases.py 3 32; ORDER=ref timeout 600 python scripts/profile_phases.py 1 32; timeout 600 python scripts/ref_order_time.py 3 1 32 256 1024 2048 4096; } > $O/ref_phases_$tag.txt 2>&1
{ timeout 600 python scripts/profile_phases.py 3 4096; timeout 600 python scripts/profile_phases.py 3 256; timeout 600 python scripts/cfg5_time.py; } > $O/phases_$tag.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_$tag.json 2> $O/bench_line_$tag.err; echo "bench rc=$?"
