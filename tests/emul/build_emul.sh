#!/bin/bash
# Build the TEST-ONLY host emulation of the HIP kernel sources (see hip_emul.h).  Never shipped.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../d3human-code_amd/csrc"
CXX=/opt/rocm/lib/llvm/bin/clang++
[ -x "$CXX" ] || CXX=clang++
OUT="$HERE/libd3h_emul.so"
$CXX -std=c++17 -O2 -ffp-contract=off -fPIC -shared -x c++ -I"$HERE/include" -I"$SRC" \
    -Wno-unused-value -Wno-pass-failed -Wno-unknown-pragmas -Wno-psabi -Wno-unused-command-line-argument \
    -o "$OUT" $(ls "$SRC"/*.hip)
echo "built $OUT"
