#!/bin/bash
# Developer script (GPU box): hardware-counter passes of the bench's solve stream, one rocprofv3 --pmc run per counter set
# (never combined with trace domains).  Output under gpurun_out/pmc_<tag>_<set>/.   scripts/pmc_passes.sh r02
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
CMD="python $R/bench.py --steps 3 --warmup 1 --no-extras --cpu-sample 0"
rocprofv3 -L > $R/gpurun_out/counters_$tag.txt 2>&1
pass() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -- $CMD > $R/gpurun_out/pmc_${tag}_$name.log 2>&1; echo "pass $name rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
pass sq2 SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $R && python - <<'PY'
import csv, glob, json, os, sys
tag = os.environ.get("TAG", "r02")
out = {}
for d in sorted(glob.glob("gpurun_out/pmc_%s_*/" % tag)):
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "solver_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        out.update(acc)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_%s_summary.json" % tag, "w"), indent=1)
PY
