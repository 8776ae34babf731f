"""bench.py, part: the live hardware counters of the value line: rocprofv3 --pmc passes of a short run of bench.py itself (HBM traffic, VALU instructions)."""
import json
import os
import sys
import time



ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from benchlib.common import (HBM_PEAK_GBS, DEV, SOLVE_FIELDS, Ctx, algorithmic_bytes, effective_cores, same_solve, same_as_ref_run,  # noqa: F401
                             bit_check)
from benchlib.stream import Stream, shard_schedule  # noqa: F401


def live_pmc(args, schedule):
    """HBM traffic and VALU instruction count of the dominant kernel, collected NOW: separate `rocprofv3 --pmc <one counter>`
    passes (nothing else enabled: no trace domain, no --stats) of a short run of this same bench -- same batch, same seed, same
    schedule, 1 warm-up + 2 timed steps -- each its own process, as MI355X_MICROARCH.md's HBM section prescribes.  Returns
    (dict or None, note).  Per batch = summed over the kernel's dispatches (queue launch + straggler launch) / queue launches."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ):
        return None, "this run is itself profiled: no nested collection"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extras", "--cpu-sample", "0",
             "--batch-per-gpu", str(args.batch_per_gpu), "--config", str(args.config), "--seed", str(args.seed), "--schedule", schedule,
             "--order", getattr(args, "order", "reference"), "--depth", str(getattr(args, "depth", 4))]
    names = ("ref4_kernel", "ref_kernel") if getattr(args, "order", "reference") == "reference" else ("solver_kernel",)
    env = dict(os.environ, TMPDIR="/tmp")
    got, t0 = {}, time.perf_counter()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        if time.perf_counter() - t0 > 150.0:
            return (got or None), "time limit reached after %s" % ", ".join(got)
        d = tempfile.mkdtemp(prefix="dftpav_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if any(nm in r["Kernel_Name"] for nm in names) and r["Counter_Name"] == ctr:
                        rows.append((int(r["Grid_Size"]), float(r["Counter_Value"]), r["Kernel_Name"]))
            if not rows:
                return (got or None), "no %s rows for %s" % (ctr, " / ".join(names))
            # per batch = everything the batch's launches counted (the queue launch and whatever follows it: the device order's
            # straggler launch, the reference order's hand-over of a batch's last trajectories to the WAVE shape) over the number of
            # batches = the launches of the PRIMARY kernel (the first of `names` that ran) with its largest grid
            prim = [nm for nm in names if any(nm in k for _, _, k in rows)][0]
            if prim == "solver_kernel":   # device order: a batch is a queue launch (the largest grid) + a straggler launch of the same kernel
                gmax = max(g for g, _, k in rows if prim in k)
                nb = sum(1 for g, _, k in rows if prim in k and g == gmax)
            else:                         # reference order: one launch of the primary kernel per batch (its grid depends on what follows it)
                nb = sum(1 for _, _, k in rows if prim in k)
            got[ctr] = sum(v for _, v, _ in rows) / nb
        except Exception as ex:  # noqa: BLE001
            return (got or None), "%s pass failed: %s" % (ctr, type(ex).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return got, "`rocprofv3 --pmc <counter> -- python bench.py %s`, one pass per counter, %.0f s" % (" ".join(child[2:]), time.perf_counter() - t0)


def hbm_traffic(ctx, args, shard_B):
    """HBM bytes per launch and VALU instructions per solve of the dominant kernel: the counters of THIS tree on THIS box
    (live_pmc) when the line carries its side runs at N = 1, else the last collection committed (profiles/pmc_latest.json).
    -> (traffic, traffic_uncorrected, traffic_source, valu_per_solve)"""
    traffic, traffic_source, valu_per_solve, traffic_raw = None, None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            if pj.get("order", "device") != getattr(args, "order", "reference"):   # a collection of the other order says nothing about this line
                raise ValueError("other order")
            traffic = pj.get("hbm_bytes_per_launch")
            valu_per_solve = pj.get("valu_instructions_per_solve")
            traffic_source = "profiles/pmc_latest.json: separate rocprofv3 --pmc passes of `%s` (%s)" % (
                pj.get("command", "bench.py --steps 3 --no-extras"), pj.get("collected", "round 1"))
        except Exception:
            traffic = None
    if ctx.world == 1 and not args.no_extras and os.environ.get("DFTPAV_BENCH_PMC", "1") != "0":
        try:
            lp, note = live_pmc(args, ctx.schedule)
        except Exception as ex:  # noqa: BLE001
            lp, note = None, "failed: %s" % type(ex).__name__
        if lp and "FETCH_SIZE" in lp and "WRITE_SIZE" in lp:
            # KB units; gfx950 counts a 128-byte read request as 64 bytes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x 2
            traffic = (2.0 * lp["FETCH_SIZE"] + lp["WRITE_SIZE"]) * 1024.0
            traffic_raw = (lp["FETCH_SIZE"] + lp["WRITE_SIZE"]) * 1024.0
            traffic_source = "live: " + note
        else:
            traffic_source = "%s [live collection: %s]" % (traffic_source, note)
        if lp and "SQ_INSTS_VALU" in lp:
            valu_per_solve = lp["SQ_INSTS_VALU"] / float(shard_B)
    return traffic, traffic_raw, traffic_source, valu_per_solve
