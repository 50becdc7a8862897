#!/bin/bash
# One rocprofv3 --pmc pass over tools/scene_bench.py with the counters named (at most 8 of the SQ block), printed
# per launch of the path kernel. Usage (inside gpurun): bash tools/pmc_pass.sh <tag> "<counters>" <scene_bench args>
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
TAG=$1; CTRS=$2; shift; shift
OUT=$REPO/gpurun_out/pmcpass_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$REPO
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p1 -- python3 $REPO/tools/scene_bench.py "$@" > $OUT/p1.log 2>&1 || { echo "pass failed"; tail -5 $OUT/p1.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
s = defaultdict(float); n = defaultdict(set)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"]:
            key = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            s[key] += float(row["Counter_Value"]); n[key].add(row["Dispatch_Id"])
for l in open(os.path.join(out, "p1.log")):
    if "kernel" in l: print(l.strip()[:110])
for k in sorted(s):
    print(f"  {k[0]:40s} {k[1]:30s} per launch {s[k] / len(n[k]):.6g}  launches {len(n[k])}")
PY
