"""Writes profiles/pmc.json, the per-world counter figures bench.py quotes beside its other_configs
(active-lane ratio, SIMD cycles per VALU wave-instruction, HBM bytes per launch), from the
per-world summaries tools/pmc_scene.sh printed and the headline summary of tools/profile.sh.
Usage: python tools/make_pmc_json.py <tag>     (reads profiles/<tag>_pmc_<world>.txt and
                                                profiles/<tag>_cornell1024_1000spp_summary.txt)"""
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
out = {}
for path in sorted(glob.glob(os.path.join(root, f"{tag}_pmc_*.txt"))):
    if "_pmc_pool_" in os.path.basename(path):
        continue  # the re-packing kernel's counters are a study, not what bench.py's configurations run on
    text = open(path).read()
    head = re.search(r"^(\S+)\s+(\d+)x(\d+) @\s*(\d+): kernel\s+([0-9.]+) ms", text, re.M)
    if not head:
        continue
    rec = {"source": os.path.basename(path), "profiled_spp": int(head.group(4)), "profiled_kernel_ms": float(head.group(5))}
    m = re.search(r"active-lane ratio .*= ([0-9.]+)", text)
    if m:
        rec["active_lane_ratio"] = float(m.group(1))
    m = re.search(r"per VALU wave-instruction = ([0-9.]+)", text)
    if m:
        rec["simd_cycles_per_valu_instruction"] = float(m.group(1))
    m = re.search(r"HBM read ([0-9.]+) MB .* written ([0-9.]+) MB per launch: ([0-9.]+) GB/s", text)
    if m:
        rec["hbm_read_mb"], rec["hbm_written_mb"], rec["hbm_gb_per_s"] = (float(m.group(i)) for i in (1, 2, 3))
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES"):
        m = re.search(rf"{k}\s+per launch ([0-9.e+]+)", text)
        if m:
            rec[k] = float(m.group(1))
    out[f"{head.group(1)} {head.group(2)}x{head.group(3)}"] = rec
headline = os.path.join(root, f"{tag}_cornell1024_1000spp_summary.txt")
if os.path.exists(headline):
    text = open(headline).read()
    rec = {"source": os.path.basename(headline), "profiled_spp": 1000}
    m = re.search(r"active-lane ratio .*= ([0-9.]+)", text)
    if m:
        rec["active_lane_ratio"] = float(m.group(1))
    m = re.search(r"SIMD cycles per VALU wave-instruction .*= ([0-9.]+)", text)
    if m:
        rec["simd_cycles_per_valu_instruction"] = float(m.group(1))
    out["cornell-box 1024x1024"] = rec
import subprocess
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                        cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
for rec in out.values():
    rec["commit"] = commit   # the tree the profiled library was built from
json.dump(out, open(os.path.join(root, "pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
