"""Static instruction-class histogram of a path kernel, phase by phase, from its ISA.

Compiles rt_kernels.hip to assembly with comment markers where the phases of the loop begin
(-DRT_PHASE_MARKS: the markers are scheduling barriers, so the listing is the shipped kernel up to
instruction order across phase borders), walks the listing of one kernel in address order and
classes every instruction by what it costs to issue (cycles per wave64 instruction measured in
profiles/r01_microbench_valu_rate*.txt). Loops are counted once (static), so the numbers say what a
phase is MADE of, not how often it runs.
Usage: python tools/isa_histogram.py [kernel-substring, default k_renderILj4168ELb0]"""
import os
import re
import subprocess
import sys
from collections import Counter, OrderedDict

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
kernel = sys.argv[1] if len(sys.argv) > 1 else "k_renderILj4168ELb0"
asm = "/tmp/rt_kernels_marks.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fno-gpu-rdc",
                       "-DRT_PHASE_MARKS", *os.environ.get("RT_EXTRA_FLAGS", "").split(), "--cuda-device-only", "-S", os.path.join(ROOT, "raytracing_amd", "csrc", "rt_kernels.hip"), "-o", asm],
                      stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN2rt\w*%s\w*:" % kernel, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def classify(op, args):
    has_sgpr = bool(re.search(r"(?<![a-z])s\d+|s\[\d+:\d+\]|vcc|exec", args.split(",", 1)[1] if "," in args else ""))
    if op.startswith("s_"):
        if op.startswith(("s_load", "s_buffer_load")): return "SMEM load"
        if op.startswith(("s_cbranch", "s_branch")): return "branch"
        if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "s_waitcnt / s_nop"
        return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM / scratch"
    if op.startswith("v_mad_u64") or op.startswith("v_mad_i64"): return "v_mad_u64_u32 (Philox multiply)"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", op): return "transcendental f32 (8.5 cyc)"
    if re.search(r"_f64", op): return "FP64 (4.5+ cyc)"
    if op.startswith("v_pk_"): return "packed f32 v_pk_* (4.5 cyc)"
    if op.startswith("v_cndmask"): return "v_cndmask (4.5 cyc)"
    if op.startswith("v_cmp"): return "v_cmp (4.5 cyc)"
    if re.match(r"v_(min|max|med)3?_", op): return "min / max / med3 (4.5 cyc)"
    if op.startswith("v_cvt"): return "convert (4.5 cyc)"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane <-> SGPR moves"
    if op.startswith(("v_mbcnt", "v_bfe", "v_lshl", "v_lshr", "v_ashr", "v_and", "v_or", "v_xor", "v_bitop3", "v_add_u32", "v_sub_u32", "v_add3", "v_mul_lo", "v_mul_hi", "v_add_co", "v_addc", "v_sub_co", "v_subrev_u32", "v_lshl_add", "v_lshl_or", "v_and_or", "v_mad_u32", "v_perm", "v_bfi", "v_not", "v_alignbit")):
        return "integer / bit ops"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
    if re.match(r"v_(fma|fmac|mul|add|sub|subrev|mac|fma_mix|fma_mixlo)_f32", op) or op.startswith("v_fma_mix"):
        return "FP32 fma/mul/add with an SGPR operand (4.5 cyc)" if has_sgpr else "FP32 fma/mul/add, VGPR operands (2.7 cyc)"
    if op.startswith("v_"): return "other VALU"
    return "other"


phases = OrderedDict()
cur = "prologue"
for l in lines[start:end]:
    m = re.search(r"; RTPHASE (\w+)", l)
    if m:
        cur = {"ts0": "queue + flush", "ts1": "camera regeneration", "ts2": "group loops", "media": "media", "record": "record (info, group ray, prim_record)",
               "unwind": "chain unwind", "ts3": "sphere draw + scatter", "ts4": "loop end / epilogue"}.get(m.group(1), m.group(1))
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", l)
    if not m or l.strip().startswith((";", ".")):
        continue
    phases.setdefault(cur, Counter())[classify(m.group(1), m.group(2))] += 1
classes = sorted({c for p in phases.values() for c in p}, key=lambda c: -sum(p[c] for p in phases.values()))
names = list(phases)
print(f"static instruction counts of {kernel}, by phase (columns) and issue class (rows)\n")
print(f"{'class':52s}" + "".join(f"{n[:14]:>15s}" for n in names) + f"{'all':>8s}")
for c in classes:
    print(f"{c:52s}" + "".join(f"{phases[n][c]:15d}" for n in names) + f"{sum(phases[n][c] for n in names):8d}")
valu = lambda p: sum(v for c, v in p.items() if not c.startswith(("SALU", "SMEM", "branch", "s_wait", "LDS", "VMEM", "other")))
print(f"{'VALU total':52s}" + "".join(f"{valu(phases[n]):15d}" for n in names) + f"{sum(valu(phases[n]) for n in names):8d}")
print(f"{'SALU + SMEM + branch':52s}" + "".join(f"{phases[n]['SALU'] + phases[n]['SMEM load'] + phases[n]['branch']:15d}" for n in names))
