"""Turns the rocprofv3 CSVs under gpurun_out/ into the tracked summaries under profiles/.

  python scripts/summarize_profiles.py r01
reads  gpurun_out/prof_<tag>/*/*_kernel_stats.csv          (rocprofv3 --kernel-trace --stats)
       gpurun_out/pmc_fetch_<tag>/*/*_counter_collection.csv (rocprofv3 --pmc FETCH_SIZE)
       gpurun_out/pmc_write_<tag>/*/*_counter_collection.csv (rocprofv3 --pmc WRITE_SIZE)
writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json, profiles/pmc_latest.json
"""
import csv, glob as _glob, json, os, shutil, sys


class glob:   # gpurun merges into existing directories: a directory may hold the files of earlier collections -- the newest only
    @staticmethod
    def glob(pattern):
        fs = sorted(_glob.glob(pattern), key=os.path.getmtime)
        return fs[-1:]
import numpy as np
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles"); os.makedirs(out, exist_ok=True)
ks = glob.glob(os.path.join(root, "gpurun_out", "prof_%s" % tag, "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(out, "%s_kernel_stats.csv" % tag))
def counter(kind):
    fs = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (kind, tag), "*", "*_counter_collection.csv"))
    vals = []
    for r in csv.DictReader(open(fs[0])):
        if "solver_kernel" in r["Kernel_Name"]:
            vals.append((int(r["Grid_Size"]), int(r["Workgroup_Size"]), float(r["Counter_Value"])))
    return vals
fetch, write = counter("fetch"), counter("write")
# One solve of a large batch is two launches of solver_kernel: the time-sliced launch (largest grid) and the
# straggler launch that follows it; "per launch" below means per solve, i.e. both of them together.
gmax = max(g for g, _, _ in fetch)
steps = sum(1 for g, _, _ in fetch if g == gmax)
wg = [t for g, t, _ in fetch if g == gmax][0]
fetch_kb, write_kb = sum(v for _, _, v in fetch) / steps, sum(v for _, _, v in write) / steps
f = [0] * steps
# MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B read
# requests at 64 B, so it is doubled before it is compared with a byte count (calibrated for 16 B/lane
# streams; our loads are 8 B/lane, so the doubled figure is an upper bound, the raw one a lower bound)
kstat = {}
if ks:
    for r in csv.DictReader(open(ks[0])):
        if "solver_kernel" in r["Name"]:
            kstat = {"rocprof_calls": int(r["Calls"]), "rocprof_sum_of_durations_ms": float(r["TotalDurationNs"]) * 1e-6}
# Kernels of consecutive batches overlap in the bench's default schedule (two streams), so the sum of their durations
# is not the time the device spent: the kernel trace gives the union of the dispatch intervals instead.  The first
# queue launch (largest grid) and whatever follows it up to the second one is the warm-up step; the rest is the timed
# region, whose union length per queue launch is what bench.py's `roofline.kernel_ms` measures with marker events.
kt = glob.glob(os.path.join(root, "gpurun_out", "prof_%s" % tag, "*", "*_kernel_trace.csv"))
if kt:
    disp = []
    for r in csv.DictReader(open(kt[0])):
        if "solver_kernel" in r["Kernel_Name"]:
            disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])))
    disp.sort()
    if disp:
        g2 = max(d[2] for d in disp)
        big = [i for i, d in enumerate(disp) if d[2] == g2]
        region = disp[big[1]:] if len(big) > 1 else disp
        nq = sum(1 for d in region if d[2] == g2)
        busy, cur_s, cur_e = 0, None, None
        for s0, e0, _ in region:
            if cur_e is None or s0 > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s0, e0
            else:
                cur_e = max(cur_e, e0)
        busy += cur_e - cur_s
        kstat.update({"rocprof_timed_queue_launches": nq, "rocprof_timeline_busy_ms": busy * 1e-6,
                      "rocprof_timeline_ms_per_batch": busy * 1e-6 / max(1, nq),
                      "rocprof_mean_queue_launch_ms": float(np.mean([(e0 - s0) * 1e-6 for s0, e0, g in region if g == g2])) if nq else None})
rec = {"tag": tag, "kernel": "solver_kernel", "solves_profiled": steps, "workgroups_time_sliced_launch": gmax // wg, "workgroup": wg,
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "hbm_bytes_per_launch_uncorrected": (fetch_kb + write_kb) * 1024.0,
       "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
       "launches_averaged": len(f)}
rec.update(kstat)
# the bench line the profiled command printed: its marker-event time per batch is the figure the timeline has to agree with
plog = os.path.join(root, "gpurun_out", "prof_%s.log" % tag)
if os.path.exists(plog):
    for line in open(plog, errors="replace"):
        k = line.find('{"metric')
        if k >= 0:
            try:
                b = json.loads(line[k:])
                rec["bench_kernel_ms_in_profiled_run"] = b["roofline"]["kernel_ms"]
                rec["bench_ms_per_step_in_profiled_run"] = b["ms_per_step"]
            except Exception:
                pass
cmdf = os.path.join(root, "gpurun_out", "profile_command_%s.txt" % tag)
rec["command"] = open(cmdf).read().strip().replace(root + "/", "") if os.path.exists(cmdf) else "python bench.py --steps 3 --no-extras"
rec["collected"] = "round %s, separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE, git %s" % (
    tag, os.popen("git -C %s rev-parse --short HEAD" % root).read().strip())
# SQ / instruction-cache / LDS counters of the same command (two more --pmc passes), summed over the solver_kernel dispatches
sq = {}
for kind in ("sq1", "sq2", "grbm"):
    for f in glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (kind, tag), "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "solver_kernel" in r["Kernel_Name"]:
                sq[r["Counter_Name"]] = sq.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
if sq:
    wc = sq.get("SQ_WAVE_CYCLES", 0.0)
    d = {"tag": tag, "command": rec["command"], "collected": rec["collected"], "batches_profiled": steps, "counters_summed_over_solver_kernel_dispatches": sq}
    if wc:
        d["fractions_of_wave_cycles"] = {k: sq[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in sq}
    if "SQC_ICACHE_REQ" in sq:
        d["icache_miss_rate"] = sq.get("SQC_ICACHE_MISSES", 0.0) / sq["SQC_ICACHE_REQ"]
    if "SQ_INSTS_VALU" in sq:
        d["valu_instructions_per_solve"] = sq["SQ_INSTS_VALU"] / steps / 4096.0
        rec["valu_instructions_per_solve"] = d["valu_instructions_per_solve"]   # bench.py: roofline.valu
    json.dump(d, open(os.path.join(out, "%s_sq.json" % tag), "w"), indent=1)
    print(json.dumps(d.get("fractions_of_wave_cycles")), d.get("icache_miss_rate"))
for src, dst in (("ref_phases_%s.txt", "%s_phases_reference_order.txt"), ("phases_%s.txt", "%s_phases.txt")):
    f_ = os.path.join(root, "gpurun_out", src % tag)
    if os.path.exists(f_):
        shutil.copy(f_, os.path.join(out, dst % tag))
kr = glob.glob(os.path.join(root, "gpurun_out", "prof_ref_%s" % tag, "*", "*_kernel_stats.csv"))
if kr:
    shutil.copy(kr[0], os.path.join(out, "%s_reference_order_kernel_stats.csv" % tag))
bl = os.path.join(root, "gpurun_out", "bench_line_%s.json" % tag)
if os.path.exists(bl):
    shutil.copy(bl, os.path.join(out, "%s_bench_line.json" % tag))
json.dump(rec, open(os.path.join(out, "%s_pmc.json" % tag), "w"), indent=1)
json.dump(rec, open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
print(json.dumps(rec))
