#!/bin/bash
# PMC passes for one scene through tools/scene_bench.py: instruction counts, waits, HBM bytes read and
# written (one rocprofv3 run per pass). Usage (inside gpurun): bash tools/pmc_scene.sh <tag> <scene> [--pool]
# Prints a summary of the path kernel per launch: active-lane ratio, SIMD cycles per VALU instruction,
# HBM GB/s (FETCH_SIZE doubled: gfx950 counts wide reads at half).
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
TAG=$1; SCENE=$2; EXTRA=${3:-}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$REPO
cd /tmp
i=1
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/tools/scene_bench.py $SCENE $EXTRA > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
  i=$((i+1))
done
python3 - $OUT $SCENE <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out, scene = sys.argv[1], sys.argv[2]
s = defaultdict(float); n = defaultdict(set); kern = set()
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"]:
            kern.add(row["Kernel_Name"])
            s[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
per = {k: s[k] / max(len(n[k]), 1) for k in s}
line = [l for l in open(os.path.join(out, "p1.log")) if l.startswith(scene)]
print(line[-1].strip() if line else scene)
print("kernel:", ", ".join(sorted(kern)))
for k in sorted(per):
    print(f"  {k:26s} per launch {per[k]:.6g}  launches {len(n[k])}")
ms = float(re.search(r"kernel\s+([0-9.]+) ms", line[-1]).group(1)) if line else 0.0
if per.get("SQ_ACTIVE_INST_VALU"):
    print(f"active-lane ratio  SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64) = {per['SQ_THREAD_CYCLES_VALU'] / (per['SQ_ACTIVE_INST_VALU'] * 64):.3f}")
if ms and per.get("SQ_INSTS_VALU"):
    print(f"SIMD cycles (2.39 GHz measured, 1024 SIMDs) per VALU wave-instruction = {ms * 1e-3 * 2.39e9 * 1024 / per['SQ_INSTS_VALU']:.2f}  (timing of the profiled run)")
    print(f"VALU : SALU : VMEM-read : LDS wave-instructions = 1 : {per['SQ_INSTS_SALU'] / per['SQ_INSTS_VALU']:.2f} : "
          f"{per.get('SQ_INSTS_VMEM_RD', 0) / per['SQ_INSTS_VALU']:.3f} : {per.get('SQ_INSTS_LDS', 0) / per['SQ_INSTS_VALU']:.3f}")
if ms and "FETCH_SIZE" in per:
    rd = per["FETCH_SIZE"] * 1024 * 2; wr = per.get("WRITE_SIZE", 0.0) * 1024
    print(f"HBM read {rd / 1e6:.1f} MB (FETCH_SIZE x2), written {wr / 1e6:.1f} MB per launch: {(rd + wr) / (ms * 1e-3) / 1e9:.1f} GB/s of about 8000")
PY
