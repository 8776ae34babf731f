#!/usr/bin/env python3
"""Developer script (no GPU): where the instructions of a kernel come from.

Compiles one translation unit of dftpav_amd/csrc again with -gline-tables-only (line tables do not change the code: the
instruction count is checked against the shipped object), disassembles the gfx950 code object, resolves every instruction of
the chosen kernel to its chain of inlined frames (llvm-symbolizer --inlines) and prints instruction counts by
  * the function called from the kernel body,
  * the source region inside chosen functions (regions = line ranges given below per function),
each split by class: fp64 arithmetic / moves, selects, cross-lane / integer VALU / SALU / LDS / global memory / scratch
(= register spills).  Static counts; along the straight-line per-round paths of the evaluation they are the dynamic counts per
round.  Kernels reached through a call (two_loop_lane, two_loop<CAP, EXACT>) are separate functions: name them.

  scripts/isa_regions.py solver      solver_kernelILb0ELi5ELi512        # device order, static obstacles only
  scripts/isa_regions.py solver      solver_kernelILb1ELi5ELi512        # device order, moving obstacles (configs[4])
  scripts/isa_regions.py solver_ref  ref_kernelILi32ELb0ELb1            # reference order, one wave per trajectory
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dftpav_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = {"solver": "-O2 -mllvm -amdgpu-sched-strategy=max-ilp", "solver_ref": "-O3 -DDFTPAV_REF_PART=1", "solver_ref_wide": "-O3 -DDFTPAV_REF_PART=2"}
COMMON = "-std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-pass-failed"

# line ranges (first, last, name) per function-name prefix: read off the `// ----` headings of the sources
REGIONS = {
    "block_eval": "solver.hip",
    "sample_point_math": "traj_math.h",
    "dynamic_pair_math": "traj_math.h",
    "ref_eval": "solver_ref.hip",
    "surround_terms": "solver_ref.hip",
    "lbfgs_advance": None,
}


def headings(path):
    """(line, text) of the `// ----` section comments of a source file: the regions"""
    out = []
    for i, ln in enumerate(open(path), 1):
        m = re.match(r"\s*// -{2,}\s*(.*?)\s*-*\s*$", ln)
        if m and m.group(1):
            out.append((i, m.group(1)[:60]))
    return out


def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_div", "v_max_f64", "v_min_f64",
                      "v_cmp", "v_ldexp", "v_frexp", "v_floor", "v_fract", "v_rndne", "v_cvt", "v_trunc", "v_ceil")):
        return "fp64"
    if op.startswith(("v_cndmask", "v_mov", "v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_swap", "v_accvgpr", "v_bfi", "v_perm")) or "dpp" in op:
        return "move"
    if op.startswith("v_"):
        return "vint"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def run(cmd, **kw):
    return subprocess.run(cmd, shell=isinstance(cmd, str), check=True, capture_output=True, text=True, **kw)


def device_elf(obj, out):
    run("%s/llvm-objcopy --dump-section .hip_fatbin=%s.fat %s" % (LLVM, out, obj))
    run("%s/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=%s.fat --output=%s" % (LLVM, out, out))


def kernel_instructions(elf, sym):
    dis = run("%s/llvm-objdump -d %s" % (LLVM, elf)).stdout
    addrs, ops, name, infn = [], [], None, False
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            infn = sym in m.group(1)
            name = m.group(1) if infn else name
            continue
        if infn:
            m = re.match(r"^\s+([a-z_0-9]+)\s.*//\s*([0-9A-F]+):", ln)
            if m:
                ops.append(m.group(1))
                addrs.append(int(m.group(2), 16))
    return name, addrs, ops


def main():
    tu, sym = sys.argv[1], sys.argv[2]
    tmp = "/tmp/isa_regions"
    os.makedirs(tmp, exist_ok=True)
    src = "solver_ref.hip" if tu.startswith("solver_ref") else tu + ".hip"
    obj = os.path.join(tmp, tu + "_g.o")
    if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))):
        run("/opt/rocm/bin/hipcc %s %s -gline-tables-only -c %s -o %s" % (COMMON, FLAGS[tu], src, obj), cwd=CSRC)
    elf = os.path.join(tmp, tu + "_g.elf")
    device_elf(obj, elf)
    name, addrs, ops = kernel_instructions(elf, sym)
    shipped = os.path.join(CSRC, tu + ".o")
    if os.path.exists(shipped):
        device_elf(shipped, os.path.join(tmp, tu + ".elf"))
        n_ship = len(kernel_instructions(os.path.join(tmp, tu + ".elf"), sym)[1])
        print("# %s: %d instructions (%d in the shipped object%s)" % (name, len(addrs), n_ship, "" if n_ship == len(addrs) else ": line tables moved %+d" % (len(addrs) - n_ship)))
    p = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + elf, "--inlines", "--output-style=JSON"], input="\n".join("0x%x" % a for a in addrs) + "\n",
                       capture_output=True, text=True, check=True)
    syms = [json.loads(ln) for ln in p.stdout.splitlines() if ln.strip()]
    heads = {f: headings(os.path.join(CSRC, f)) for f in ("solver.hip", "solver_ref.hip", "traj_math.h")}

    def region(fname, line):
        hs = heads.get(os.path.basename(fname), [])
        cur = "(top)"
        for ln, text in hs:
            if ln <= line:
                cur = text
            else:
                break
        return cur

    def short(fn):
        fn = fn.replace("dftpav::", "")
        fn = re.sub(r"const __attribute__\(\(address_space\((\d)\)\)\) double \*", r"as\1", fn)
        return re.sub(r"\(.*$", "", fn)[:64]

    top, inner = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
    for op, s in zip(ops, syms):
        c = classify(op)
        fr = s.get("Symbol", [])
        names = [short(f["FunctionName"]) for f in fr]
        top[names[-2] if len(names) >= 2 else "(kernel body)"][c] += 1
        for f in fr:
            base = short(f["FunctionName"]).split("<")[0]
            if base in REGIONS:
                inner[(short(f["FunctionName"]), region(f["FileName"], f["Line"]))][c] += 1
    order = ("fp64", "move", "vint", "salu", "lds", "vmem", "scratch", "other")
    print("# classes: " + " ".join(order))
    print("## by function called from the kernel body")
    for k, v in sorted(top.items(), key=lambda kv: -sum(kv[1].values())):
        print("%-66s %6d   %s" % (k, sum(v.values()), " ".join("%5d" % v[c] for c in order)))
    print("## by source region inside the evaluators (the `// ----` headings of the sources)")
    for k, v in sorted(inner.items(), key=lambda kv: (kv[0][0], -sum(kv[1].values()))):
        print("%-44s %-62s %6d   %s" % (k[0][:44], k[1], sum(v.values()), " ".join("%5d" % v[c] for c in order)))


if __name__ == "__main__":
    main()
