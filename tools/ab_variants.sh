#!/bin/bash
# Dev tool: build variants of the library with -D flags for same-box A/B timing:
#   tools/ab_variants.sh name1:"-DFLAG1 -DFLAG2" name2:"-DFLAG3" ...  ->  latok_amd/liblatok_hip_<name>.so
# then on the GPU box: LATOK_HIP_LIB=$PWD/latok_amd/liblatok_hip_<name>.so python tools/quick_bench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  TMP=$(mktemp -d)
  mkdir -p "$TMP/latok_amd" "$TMP/include"
  cp -r "$ROOT/latok_amd/csrc" "$TMP/latok_amd/csrc"
  cp "$ROOT/include/latok_hip.h" "$TMP/include/"
  rm -f "$TMP"/latok_amd/csrc/*.o
  make -s -C "$TMP/latok_amd/csrc" FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags" >/dev/null
  cp "$TMP/latok_amd/liblatok_hip.so" "$ROOT/latok_amd/liblatok_hip_$name.so"
  rm -rf "$TMP"
  echo "built $name ($flags)"
done
