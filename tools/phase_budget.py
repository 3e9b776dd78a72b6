#!/usr/bin/env python3
"""Per-phase instruction budget of a kernel from its gfx950 assembly (no GPU needed).

    python3 tools/phase_budget.py [--file fft4096_features.hip] [--kernel fft4096_features_kernelILb1] [--loops]

Compiles the file with -DSDRK_PHASE_MARKS -S (device side only): SDRK_PHASE("name") (kernels.h) leaves a comment in the
assembly wherever the source enters a phase.  The instruction stream of the kernel is cut at those comments IN LAYOUT
ORDER and counted by class — VALU (v_*), SALU (s_* except waits / branches / barriers), LDS (ds_*), VMEM (buffer_* /
global_* / scratch_*), waits (s_waitcnt, s_barrier, s_nop), branches.  These are STATIC counts: a loop body counts once
(--loops lists every backward branch with the size of its body and the phase it lies in, so that trip counts can be put
against the counters' dynamic totals by hand), a block the compiler laid out elsewhere (a cold path) counts where it
lies.  The marked build differs slightly from the shipped one (the markers fence the scheduler)."""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdr-iq-visualizer_amd", "csrc")
ap = argparse.ArgumentParser()
ap.add_argument("--file", default="fft4096_features.hip")
ap.add_argument("--kernel", default="fft4096_features_kernelILb1")
ap.add_argument("--loops", action="store_true")
ap.add_argument("--asm", default="/tmp/phase_budget.s")
a = ap.parse_args()
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-ffp-contract=on",
       "-DSDRK_PHASE_MARKS", "-S", "--cuda-device-only", "-o", a.asm, os.path.join(CSRC, a.file)]
r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
if r.returncode != 0:
    sys.exit(r.stderr[-2000:])
lines = open(a.asm).read().splitlines()
start = next(i for i, ln in enumerate(lines) if ln.startswith("_ZN") and a.kernel in ln and ln.rstrip().split(":")[0].endswith(ln.split(":")[0]))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def klass(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "VMEM"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    return "other"


phase = "prologue"
order, counts = ["prologue"], collections.defaultdict(collections.Counter)
labels, instr_index, n_instr, loops = {}, [], 0, []
for ln in lines[start + 1:end]:
    t = ln.strip()
    m = re.match(r";\s*SDRK_PHASE (\w+)", t)
    if m:
        phase = m.group(1)
        if phase not in order:
            order.append(phase)
        continue
    m = re.match(r"(\.LBB\d+_\d+):", t)
    if m:
        labels[m.group(1)] = (n_instr, phase)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    counts[phase][klass(op)] += 1
    n_instr += 1
    if op.startswith(("s_cbranch", "s_branch")):
        tgt = t.split()[-1]
        if tgt in labels:                                     # backward branch: a loop
            loops.append((labels[tgt][1], phase, tgt, n_instr - labels[tgt][0]))
cols = ["VALU", "SALU", "LDS", "VMEM", "wait", "branch", "other"]
print(f"# {a.file} :: {a.kernel}  —  static instructions per phase (layout order), gfx950, -DSDRK_PHASE_MARKS build")
print("%-34s" % "phase" + "".join("%8s" % c for c in cols) + "%8s" % "all")
tot = collections.Counter()
for ph in order:
    c = counts[ph]
    tot.update(c)
    print("%-34s" % ph + "".join("%8d" % c[k] for k in cols) + "%8d" % sum(c.values()))
print("%-34s" % "total" + "".join("%8d" % tot[k] for k in cols) + "%8d" % sum(tot.values()))
if a.loops:
    print("\n# backward branches (loops): phase of the loop head -> phase of the branch, target label, static body size")
    for head, tail, tgt, size in loops:
        print(f"  {head:32s} -> {tail:32s} {tgt:12s} {size:6d} instructions")
