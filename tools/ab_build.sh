#!/bin/bash
# Dev tool: build the library of another git revision as latok_amd/liblatok_hip_ab.so for same-box A/B timing
#   tools/ab_build.sh HEAD~1 ; then on the GPU box: LATOK_HIP_LIB=$PWD/latok_amd/liblatok_hip_ab.so python tools/quick_bench.py
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" latok_amd/csrc include | tar -x -C "$TMP"
make -s -C "$TMP/latok_amd/csrc" >/dev/null
cp "$TMP/latok_amd/liblatok_hip.so" "$ROOT/latok_amd/liblatok_hip_ab.so"
rm -rf "$TMP"
echo "built $REV -> latok_amd/liblatok_hip_ab.so"
