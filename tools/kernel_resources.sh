#!/bin/bash
# Dev tool: VGPRs / scratch / LDS / occupancy of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)
#   tools/kernel_resources.sh latok_amd/csrc/split_kernels.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  sed 's/ \[-Rpass.*//' |
  awk '/Function Name:/ {n=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {l=$NF; printf "%s VGPR %s scratch %s occ %s LDS %s\n", n, v, s, o, l}' | c++filt | sed 's/latok:://; s/(latok::[A-Za-z]*)//' | sort
