"""The compiler's resource report for every kernel of librtmi.so (registers, spills, scratch, waves per
SIMD the registers allow): hipcc -Rpass-analysis=kernel-resource-usage on the sources, as a table.
Usage: python tools/kernel_resources.py > profiles/<tag>_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "raytracing_amd", "csrc")
rows = []
for f in ("rt_kernels.hip", "rt_bvh.hip", "rt_api.hip", "rt_multi.hip"):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-fno-gpu-rdc",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(SRC, f), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark: (Function Name|\s+[A-Za-z ]+(?:\[[^\]]*\])?): (.*) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} {'VGPRs':>6s} {'AGPRs':>6s} {'SGPRs':>6s} {'VGPR spill':>10s} {'SGPR spill':>10s} {'scratch B/lane':>14s} {'waves/SIMD':>10s} {'static LDS':>10s}")
for r, d in zip(rows, demangle):
    d = re.sub(r"\(rt::RenderArgs\)|\(.*\)$", "", d)[:70]
    print(f"{d:70s} {r.get('VGPRs', '?'):>6s} {r.get('AGPRs', '?'):>6s} {r.get('SGPRs', r.get('TotalSGPRs', '?')):>6s} {r.get('VGPRs Spill', '?'):>10s} {r.get('SGPRs Spill', '?'):>10s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>14s} {r.get('Occupancy [waves/SIMD]', '?'):>10s} {r.get('LDS Size [bytes/block]', '?'):>10s}")
print("\n(k_render<F, false> instantiations by mask: 72 = V_RECT, 76 = V_SMOKE, 35 = V_BALLS (see rt_kernels.hip for the other masks), the counting build of V_ALL_DEEP and the parity hooks;"
      "\n dynamic LDS is asked for at launch and is not in the last column: rt_scene_get_info().lds_bytes)")
