#!/bin/bash
# Registers / spills of the k_render instantiations of rt_kernels.hip with extra flags: bash tools/kres.sh [-D...]
cd "$(dirname "$0")/../raytracing_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc "$@" \
  -Rpass-analysis=kernel-resource-usage -c rt_kernels.hip -o /dev/null 2>&1 | \
  grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - | grep "k_render" | c++filt | sed 's/(rt::RenderArgs)//; s/Function Name: void rt:://; s/\[waves\/SIMD\]//'
