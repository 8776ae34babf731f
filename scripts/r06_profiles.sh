#!/bin/bash
# GPU box, round 6: what profiles/r06_* is summarised from.  rocprofv3 runs from /tmp with TMPDIR=/tmp; the counter passes are separate
# runs with --pmc only (never combined with a trace domain).   scripts/r06_profiles.sh ; then here: python scripts/r06_summarize.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0"
echo "$CMD" > $O/r06_profile_command.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_prof -- $CMD > $O/r06_prof.log 2>&1; echo "kernel trace rc=$?"
pass() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/r06_pmc_$name -- $CMD > $O/r06_pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
pass sq2 SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
# the reference-order kernels of the other BASELINE configurations: configs[4] at 1024 (moving obstacles, ref_kernel<64,true,true>) and
# configs[1] at 4096 (gear shift, ref_kernel<40,false,true>): kernel statistics and counters
C5="python $R/scripts/ref_order_time.py 5 1024"
C2="python $R/scripts/ref_order_time.py 2 4096"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_prof_cfg5 -- $C5 > $O/r06_prof_cfg5.log 2>&1; echo "cfg5 trace rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_prof_cfg2 -- $C2 > $O/r06_prof_cfg2.log 2>&1; echo "cfg2 trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/r06_pmc_cfg5_$c -- $C5 > $O/r06_pmc_cfg5_$c.log 2>&1; echo "cfg5 $c rc=$?"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/r06_pmc_cfg5_sq -- $C5 > $O/r06_pmc_cfg5_sq.log 2>&1; echo "cfg5 sq rc=$?"
cd $R
{ ORDER=ref timeout 600 python scripts/profile_phases.py 3 4096; ORDER=ref DFTPAV_REF_SHAPE=quad timeout 300 python scripts/profile_phases.py 3 8; ORDER=ref timeout 600 python scripts/profile_phases.py 2 4096; ORDER=ref timeout 900 python scripts/profile_phases.py 5 1024; for c in 2 3; do CFG=$c timeout 300 python scripts/ref_stream_time.py 2 4 8; CFG=$c timeout 300 python scripts/stream_profile.py 4; done; } > $O/r06_phases_reference_order.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_bench_line.json 2> $O/r06_bench_line.err; echo "bench rc=$?"
tail -c 400 $O/r06_bench_line.err
