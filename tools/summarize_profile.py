"""Condenses a tools/profile.sh output directory into a text summary (kernel stats + PMC sums
for the path kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("  ", {k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== per-dispatch (kernel trace): k_render<false> launches")
durations = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "k_render" in row.get("Kernel_Name", "") and "false>" in row.get("Kernel_Name", ""):
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                durations.append(dur)
                # The trace's register and LDS columns as rocprofv3 gives them, named as such: on gfx950 its
                # VGPR_Count came out as HALF the per-lane allocation the compiler reports (36 for a 72-register
                # kernel) and LDS_Block_Size leaves out dynamic LDS (0 for a kernel that asks for its LDS at
                # launch). The compiler's own figures are in profiles/*_kernel_resources.txt.
                extra = {k: row[k] for k in row if any(t in k for t in ("LDS", "Shared", "Scratch", "Private")) and row[k] not in ("", None)}
                print("  ", row["Kernel_Name"][:60], "ms", round(dur, 3), "| rocprofv3 trace columns: VGPR_Count", row.get("VGPR_Count"),
                      "Accum_VGPR_Count", row.get("Accum_VGPR_Count"), "SGPR_Count", row.get("SGPR_Count"), extra,
                      "grid", row.get("Grid_Size_X", row.get("Grid_Size")), "workgroup", row.get("Workgroup_Size_X", row.get("Workgroup_Size")))
print("== PMC counters summed over the k_render<false> dispatches (per launch = sum / launches)")
sums = defaultdict(float)
launches = defaultdict(set)
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            kn = row.get("Kernel_Name", "")
            if "k_render" in kn and "false>" in kn:
                sums[row["Counter_Name"]] += float(row["Counter_Value"])
                launches[row["Counter_Name"]].add(row["Dispatch_Id"])
for k in sorted(sums):
    n = max(len(launches[k]), 1)
    print(f"   {k:28s} total {sums[k]:.6g}  launches {n}  per launch {sums[k] / n:.6g}")
g = lambda k: sums.get(k, 0.0) / max(len(launches.get(k, [1])), 1)
if g("SQ_INSTS_VALU"):
    print("== derived (per launch)")
    print(f"   active-lane ratio  SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64) = {g('SQ_THREAD_CYCLES_VALU') / max(g('SQ_ACTIVE_INST_VALU') * 64, 1):.4f}")
    if durations:
        # SIMD cycles available per VALU wave-instruction: kernel time x shader clock x SIMDs / instructions
        # (2.39 GHz is what one wave measures inside the kernel, rt_stats.shader_mhz; 256 CUs x 4 SIMDs)
        ms = sum(durations) / len(durations)
        print(f"   SIMD cycles per VALU wave-instruction  {ms:.3f} ms x 2.39e9 x 1024 / SQ_INSTS_VALU  = {ms * 1e-3 * 2.39e9 * 1024 / g('SQ_INSTS_VALU'):.3f}")
        print(f"   SALU : VALU wave-instructions = {g('SQ_INSTS_SALU') / g('SQ_INSTS_VALU'):.3f}")
    print(f"   waves {g('SQ_WAVES'):.0f}, VALU insts/wave {g('SQ_INSTS_VALU') / max(g('SQ_WAVES'), 1):.0f}, SALU insts/wave {g('SQ_INSTS_SALU') / max(g('SQ_WAVES'), 1):.0f}")
if g("FETCH_SIZE") or g("WRITE_SIZE"):
    print(f"   HBM read  (FETCH_SIZE KiB, gfx950 wide-read correction x2 NOT applied): {g('FETCH_SIZE'):.6g} KiB")
    print(f"   HBM write (WRITE_SIZE KiB): {g('WRITE_SIZE'):.6g} KiB")
