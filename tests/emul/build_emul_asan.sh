#!/bin/bash
# The host emulation of the kernel sources with AddressSanitizer + UBSan (sanitizers belong on the CPU build: GPU ASan is unavailable on the
# pool).  Loaded into python with the sanitizer runtime preloaded -- tools/dbg/asan_tick.sh.  Globals (the emulated __shared__ arrays) and
# every heap block torch hands to a kernel get red zones: an out-of-bounds LDS or global access of any kernel aborts with a report.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../d3human-code_amd/csrc"
CXX=/opt/rocm/lib/llvm/bin/clang++
OUT="$HERE/libd3h_emul_asan.so"
$CXX -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize=vptr,function -shared-libsan -ffp-contract=off -fPIC -shared -x c++ \
    -I"$HERE/include" -I"$SRC" -Wno-unused-value -Wno-pass-failed -Wno-unknown-pragmas -Wno-psabi -Wno-unused-command-line-argument \
    -o "$OUT" $(ls "$SRC"/*.hip)
echo "built $OUT"
