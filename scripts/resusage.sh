#!/bin/bash
# Per-kernel register / LDS / spill summary of one .hip file (cross-compiles, no GPU needed).
#   scripts/resusage.sh wakeword-jupyterlab_amd/csrc/ww_cnn.hip [extra hipcc flags]
f=$1; shift
cd "$(dirname "$f")" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -DWW_BUILD -fno-slp-vectorize \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$(basename "$f")" -o /dev/null 2>&1 | grep "remark:" |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /TotalSGPRs:/ {sg=$NF} /VGPRs Spill/ {sp=$NF} /LDS Size/ {l=$NF; printf "%-72s vgpr %3s agpr %3s sgpr %3s scratch %4s spill %3s occ %s lds %s\n", name, v, a, sg, s, sp, o, l}' | c++filt | sed -E "s/\(float const.*\)  +vgpr/(...) vgpr/" | cut -c1-200
