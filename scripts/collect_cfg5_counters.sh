#!/bin/bash
# GPU box: HBM traffic and SQ counters of the moving-obstacle kernel solver_kernel<true,...> on BASELINE configs[4] (batch 1024),
# separate rocprofv3 --pmc passes; summarised by the python below into gpurun_out/cfg5_counters_<tag>.json
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/cfg5_time.py 1024"
pass() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/cfg5_${name}_$tag -- $CMD > $O/cfg5_${name}_$tag.log 2>&1; echo "cfg5 $name rc=$?"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
pass mix SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd $R
python - <<PY
import csv, glob, json, collections, os
out = {"tag": "$tag", "kernel": "solver_kernel<true,6,512> (device order, moving obstacles), BASELINE configs[4], batch 1024", "command": "scripts/cfg5_time.py 1024 (4 solves of the batch per pass)"}
tot = collections.defaultdict(float); nd = 0
for name in ("fetch", "write", "sq1", "mix"):
    fs = sorted(glob.glob("$O/cfg5_%s_$tag/*/*_counter_collection.csv" % name), key=os.path.getmtime)
    if not fs: continue
    disp = set()
    for r in csv.DictReader(open(fs[-1])):
        if "solver_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    nd = max(nd, len(disp))
solves = 4 * 1024.0
out["dispatches"] = nd
out["per_solve"] = {k: v / solves for k, v in tot.items()}
if "FETCH_SIZE" in tot:
    out["hbm_bytes_per_batch_uncorrected"] = (tot["FETCH_SIZE"] + tot.get("WRITE_SIZE", 0.0)) * 1024.0 / 4
    out["hbm_bytes_per_batch_fetch_doubled"] = (2 * tot["FETCH_SIZE"] + tot.get("WRITE_SIZE", 0.0)) * 1024.0 / 4
if tot.get("SQ_WAVE_CYCLES"):
    out["fractions_of_wave_cycles"] = {k: tot[k] / tot["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in tot}
json.dump(out, open("$O/cfg5_counters_$tag.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
