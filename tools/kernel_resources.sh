#!/bin/bash
# Register / LDS / spill report of the kernels of one csrc/*.hip file (compiler remarks, no GPU needed):
#   tools/kernel_resources.sh speech2affective_gestures_amd/csrc/wave_fused.hip [name-filter]
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/speech2affective_gestures_amd/csrc \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | grep "remark:" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | \
  awk -F': ' '/Function Name/{name=$2} /^VGPRs:/{v=$2} /^AGPRs/{a=$2} /TotalSGPRs/{sg=$2} /ScratchSize/{sc=$2} /VGPRs Spill/{sp=$2} /Occupancy/{o=$2} /LDS Size/{printf "%-90s vgpr %3s agpr %3s sgpr %3s scratch %4s spill %3s occ %s lds %s\n", name, v, a, sg, sc, sp, o, $2}' | \
  (command -v c++filt >/dev/null && c++filt || cat) | grep -E "${2:-.}"
