#!/bin/bash
# PMC pass (instruction counts) for one scene through tools/scene_bench.py. Usage (inside gpurun): bash tools/pmc_scene.sh <tag> <scene>
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
TAG=$1; SCENE=$2
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$REPO
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/p1 -- python3 $REPO/tools/scene_bench.py $SCENE > $OUT/p1.log 2>&1 || { echo "pass 1 failed"; tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- python3 $REPO/tools/scene_bench.py $SCENE > $OUT/p2.log 2>&1 || { echo "pass 2 failed"; tail -5 $OUT/p2.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
s = defaultdict(float); n = defaultdict(set)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"]:
            s[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
for k in sorted(s):
    print(f"{k:26s} per launch {s[k] / max(len(n[k]), 1):.6g}  launches {len(n[k])}")
PY
grep -v amdgpu.ids $OUT/p1.log | tail -3
