#!/bin/bash
# Instruction-mix PMC passes for bench.py (one rocprofv3 run each). Usage (inside gpurun): bash tools/pmc_mix.sh <tag> [bench args]
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
TAG=${1:-mix}; shift || true
ARGS=${@:-"--steps 1 --warmup 1 --no-cpu-baseline"}
OUT=$REPO/gpurun_out/mix_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_BRANCH SQ_INSTS_SALU"
i=1
for P in "$P1" "$P2"; do
  timeout -s KILL 300 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
s = defaultdict(float); n = defaultdict(set)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"] and "false>" in row["Kernel_Name"]:
            s[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
print("per launch of k_render<..., false> (wave-instruction counts unless the counter says FLOPS):")
for k in sorted(s):
    print(f"  {k:28s} {s[k] / max(len(n[k]), 1):.6g}  (launches {len(n[k])})")
PY
