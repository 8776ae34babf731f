"""Turns the rocprofv3 CSVs of scripts/r06_profiles.sh (gpurun_out/r06_*) into the tracked summaries under profiles/:
   r06_kernel_stats.csv (the profiled bench command), r06_kernel_stats_cfg2.csv / _cfg5.csv (the reference-order kernels of the other
   BASELINE configurations), r06_pmc.json (counters per batch of the value line's kernel), r06_pmc_cfg5.json, pmc_latest.json.
   python scripts/r06_summarize.py"""
import csv
import glob
import json
import os
import shutil

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


for src, dst in (("r06_prof", "r06_kernel_stats.csv"), ("r06_prof_cfg2", "r06_kernel_stats_cfg2.csv"), ("r06_prof_cfg5", "r06_kernel_stats_cfg5.csv")):
    f = newest(os.path.join(G, src, "*", "*_kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(P, dst))


def counters(dirs, names):
    out = {}
    for d in dirs:
        f = newest(os.path.join(G, d, "*", "*_counter_collection.csv"))
        if not f:
            continue
        rows = [r for r in csv.DictReader(open(f)) if any(nm in r["Kernel_Name"] for nm in names)]
        for c in sorted(set(r["Counter_Name"] for r in rows)):
            rr = [(int(r["Grid_Size"]), float(r["Counter_Value"]), r["Kernel_Name"]) for r in rows if r["Counter_Name"] == c]
            prim = [nm for nm in names if any(nm in k for _, _, k in rr)][0]
            nb = sum(1 for _, _, k in rr if prim in k)   # one launch of the primary kernel per batch
            out[c] = {"per_batch": sum(v for _, v, _ in rr) / nb, "batches": nb, "dispatches": len(rr)}
    return out


cmd = open(os.path.join(G, "r06_profile_command.txt")).read().strip() if os.path.exists(os.path.join(G, "r06_profile_command.txt")) else ""
def dispatch_union():
    """the profiled bench command's timed region in its kernel trace: union of the dispatch intervals of the solve kernels launched by the 4 timed
    steps (the launches of a stream overlap: their average duration is not a per-batch time) / 4, beside that run's own ms_per_step"""
    f = newest(os.path.join(G, "r06_prof", "*", "*_kernel_trace.csv"))
    if not f:
        return None
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))
            if "ref4_kernel" in r["Kernel_Name"] or "ref_kernel" in r["Kernel_Name"]]
    rows.sort()
    quad = [r for r in rows if "ref4_kernel" in r[2]]
    if len(quad) < 5:
        return None
    t_first = quad[-4][0]                      # 1 warm-up step (a batch alone), then 4 timed steps
    timed = sorted((a, b) for a, b, _ in rows if a >= t_first)
    total, cur_a, cur_b = 0, timed[0][0], timed[0][1]
    for a, b in timed[1:]:
        if a > cur_b:
            total += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    total += cur_b - cur_a
    out = {"dispatch_union_ms_per_batch": total / 4 / 1e6, "launches_in_the_timed_region": len(timed)}
    log = os.path.join(G, "r06_prof.log")
    if os.path.exists(log):
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
        if lines:
            out["ms_per_step_of_that_run_by_its_own_clock"] = json.loads(lines[-1])["ms_per_step"]
    return out


c = counters(["r06_pmc_fetch", "r06_pmc_write", "r06_pmc_sq1", "r06_pmc_sq2"], ("ref4_kernel", "ref_kernel"))
B = 4096
if c:
    fetch, write = c["FETCH_SIZE"]["per_batch"] * 1024.0, c["WRITE_SIZE"]["per_batch"] * 1024.0
    j = {"order": "reference", "kernel": "ref4_kernel<true> (QUAD shape) + the WAVE-shape launch that finishes the last batch's last trajectories",
         "command": cmd, "collected": "round 6", "batch": B,
         "hbm_bytes_per_launch": 2.0 * fetch + write, "hbm_bytes_per_launch_uncorrected": fetch + write,
         "note": "FETCH_SIZE / WRITE_SIZE in KB; gfx950 counts a 128-byte read request as 64 bytes: FETCH doubled (MI355X_MICROARCH.md, HBM section); per batch = all "
                 "dispatches of the profiled run / launches of the primary kernel",
         "valu_instructions_per_solve": c["SQ_INSTS_VALU"]["per_batch"] / B,
         "counters_per_batch": {k: v["per_batch"] for k, v in c.items()},
         "derived": {"valu_active_fraction_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"]["per_batch"] / c["SQ_WAVE_CYCLES"]["per_batch"],
                     "waiting_fraction_of_wave_cycles": c["SQ_WAIT_ANY"]["per_batch"] / c["SQ_WAVE_CYCLES"]["per_batch"],
                     "icache_miss_rate": c["SQC_ICACHE_MISSES"]["per_batch"] / c["SQC_ICACHE_REQ"]["per_batch"],
                     "salu_per_valu": c["SQ_INSTS_SALU"]["per_batch"] / c["SQ_INSTS_VALU"]["per_batch"],
                     "lds_instructions_per_solve": c["SQ_INSTS_LDS"]["per_batch"] / B, "vmem_reads_per_solve": c["SQ_INSTS_VMEM_RD"]["per_batch"] / B}}
    json.dump(j, open(os.path.join(P, "r06_pmc.json"), "w"), indent=1)
    json.dump(j, open(os.path.join(P, "pmc_latest.json"), "w"), indent=1)
    du = dispatch_union()
    if du:
        j["kernel_trace_of_the_same_command"] = du
        json.dump(j, open(os.path.join(P, "r06_pmc.json"), "w"), indent=1)
        print("r06_pmc.json: kernel trace", du)
    print("r06_pmc.json: traffic %.1f GB corrected / %.1f raw, %.2f M VALU per solve, VALU active %.2f, waiting %.2f" % (
        j["hbm_bytes_per_launch"] / 1e9, j["hbm_bytes_per_launch_uncorrected"] / 1e9, j["valu_instructions_per_solve"] / 1e6,
        j["derived"]["valu_active_fraction_of_wave_cycles"], j["derived"]["waiting_fraction_of_wave_cycles"]))
c5 = counters(["r06_pmc_cfg5_FETCH_SIZE", "r06_pmc_cfg5_WRITE_SIZE", "r06_pmc_cfg5_sq"], ("ref_kernel",))
if c5:
    j5 = {"kernel": "ref_kernel<64, SUR, *> on BASELINE configs[4] at batch 1024 (reference order, moving obstacles)", "batch": 1024,
          "command": "python scripts/ref_order_time.py 5 1024 (three solves: per batch = per solve of the batch)",
          "counters_per_batch": {k: v["per_batch"] for k, v in c5.items()}}
    if "FETCH_SIZE" in c5 and "WRITE_SIZE" in c5:
        j5["hbm_bytes_per_batch"] = (2.0 * c5["FETCH_SIZE"]["per_batch"] + c5["WRITE_SIZE"]["per_batch"]) * 1024.0
    if "SQ_INSTS_VALU" in c5:
        j5["valu_instructions_per_solve"] = c5["SQ_INSTS_VALU"]["per_batch"] / 1024
        if "SQ_WAVE_CYCLES" in c5:
            j5["valu_active_fraction_of_wave_cycles"] = c5["SQ_ACTIVE_INST_VALU"]["per_batch"] / c5["SQ_WAVE_CYCLES"]["per_batch"]
            j5["waiting_fraction_of_wave_cycles"] = c5["SQ_WAIT_ANY"]["per_batch"] / c5["SQ_WAVE_CYCLES"]["per_batch"]
    json.dump(j5, open(os.path.join(P, "r06_pmc_cfg5.json"), "w"), indent=1)
    print("r06_pmc_cfg5.json:", {k: v for k, v in j5.items() if k not in ("counters_per_batch", "kernel", "command")})
for f, dst in (("r06_phases_reference_order.txt", "r06_phases_reference_order.txt"), ("r06_bench_line.json", "r06_bench_line.json")):
    if os.path.exists(os.path.join(G, f)):
        shutil.copy(os.path.join(G, f), os.path.join(P, dst))
