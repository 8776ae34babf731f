"""Turns the SQ_INSTS_* counter passes of scripts/collect_instmix.sh into profiles/<tag>_instmix.json: the DYNAMIC instruction mix
by class of the device-order kernel solver_kernel<false,5,512> and of the reference-order kernel ref_kernel<32,false,true>, per
solve.  (gfx950 in this image has neither PC sampling nor a thread-trace decoder; the per-class counters are the dynamic
instruction mix the hardware offers.)  python scripts/summarize_instmix.py r04"""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(kind, name, kernel):
    fs = sorted(glob.glob(os.path.join(root, "gpurun_out", "mix_%s_%s_%s" % (kind, name, tag), "*", "*_counter_collection.csv")), key=os.path.getmtime)
    acc, disp = collections.defaultdict(float), collections.defaultdict(set)
    if not fs:
        return {}, 0
    for r in csv.DictReader(open(fs[-1])):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    n = max((len(v) for v in disp.values()), default=0)
    return dict(acc), n


out = {"tag": tag, "method": "rocprofv3 --pmc, one pass per counter group (scripts/collect_instmix.sh); counts are wave-level instructions "
       "summed over the kernel's dispatches of the profiled command and divided by the solves those dispatches ran",
       "why_not_pc_sampling": "rocprofv3 --pc-sampling-* on this image / gfx950: 'Given PC sampling configuration is not supported on any of the "
                              "agents'; no thread-trace decoder library under /opt/rocm/lib"}
for kind, kernel, solves_per_dispatch_set, label in (("dev", "solver_kernel", None, "solver_kernel<false,5,512> (device order)"),
                                                      ("ref", "ref_kernel", None, "ref_kernel<32,false,true> (reference order, WAVE shape)")):
    tot = {}
    nd = 0
    for name in ("a", "b", "c"):
        c, n = counters(kind, name, kernel)
        tot.update(c)
        nd = max(nd, n)
    if not tot:
        continue
    # dev: bench.py --steps 4 --warmup 1 -> 5 batches of 4096 in the queue launches (+ the list launches of the same solves);
    # ref: ref_order_time.py 3 4096 -> 3 reference-order solves of 4096 (the device-order solves of that script are solver_kernel)
    solves = 5 * 4096 if kind == "dev" else 3 * 4096
    per = {k: v / solves for k, v in tot.items()}
    valu = per.get("SQ_INSTS_VALU", 0.0)
    f64 = sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    f32 = sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"))
    integer = per.get("SQ_INSTS_VALU_INT32", 0.0) + per.get("SQ_INSTS_VALU_INT64", 0.0)
    cvt = per.get("SQ_INSTS_VALU_CVT", 0.0)
    other = valu - f64 - f32 - integer - cvt
    rec = {"kernel": label, "dispatches_profiled": nd, "solves": solves, "per_solve": {k: round(v, 1) for k, v in sorted(per.items())},
           "valu_mix_per_solve": {"fp64_arithmetic (add + mul + fma + trans)": round(f64), "fp32_arithmetic": round(f32),
                                  "integer (int32 + int64: addresses, masks, counters)": round(integer), "conversions": round(cvt),
                                  "rest (v_mov, v_cndmask, compares, DPP / permlane / readlane, ...)": round(other)},
           "valu_mix_fraction": {"fp64_arithmetic": f64 / valu if valu else None, "integer": integer / valu if valu else None,
                                 "conversions": cvt / valu if valu else None, "rest": other / valu if valu else None},
           "non_valu_per_solve": {k: round(per.get(k, 0.0)) for k in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_LDS_STORE",
                                                                     "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")}}
    if "SQ_WAVE_CYCLES" in tot:
        rec["wave_cycles"] = {"wait_any_frac": tot.get("SQ_WAIT_ANY", 0) / tot["SQ_WAVE_CYCLES"], "valu_active_frac": tot.get("SQ_ACTIVE_INST_VALU", 0) / tot["SQ_WAVE_CYCLES"],
                              "any_inst_active_frac": tot.get("SQ_ACTIVE_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"]}
    out[kind] = rec
# reference-order kernel: HBM traffic and the other SQ counters of its own passes
for name in ("fetch", "write", "sq2"):
    c, n = counters("ref", name, "ref_kernel")
    if c:
        out.setdefault("ref_counters", {}).update({k: v / max(1, n) for k, v in c.items()})
        out["ref_counters"]["dispatches"] = n
if "ref_counters" in out:
    rc = out["ref_counters"]
    if "FETCH_SIZE" in rc and "WRITE_SIZE" in rc:
        rc["hbm_bytes_per_batch_uncorrected"] = (rc["FETCH_SIZE"] + rc["WRITE_SIZE"]) * 1024.0
        rc["hbm_bytes_per_batch_fetch_doubled"] = (2 * rc["FETCH_SIZE"] + rc["WRITE_SIZE"]) * 1024.0
    if rc.get("SQC_ICACHE_REQ"):
        rc["icache_miss_rate"] = rc.get("SQC_ICACHE_MISSES", 0.0) / rc["SQC_ICACHE_REQ"]
json.dump(out, open(os.path.join(root, "profiles", "%s_instmix.json" % tag), "w"), indent=1)
print(json.dumps({k: (v.get("valu_mix_per_solve"), v.get("valu_mix_fraction"), v.get("non_valu_per_solve"), v.get("wave_cycles")) if isinstance(v, dict) and "valu_mix_per_solve" in v else None for k, v in out.items() if k in ("dev", "ref")}, indent=1))
print(out.get("ref_counters"))
