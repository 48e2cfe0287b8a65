#!/bin/bash
# bash tools/dbg/asan_tick.sh [init full split seq]   (CPU container; builds tests/emul/libd3h_emul_asan.so when missing)
cd "$(dirname "$0")/../.."
[ -f tests/emul/libd3h_emul_asan.so ] || tests/emul/build_emul_asan.sh
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0:allocator_may_return_null=1:handle_segv=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0 python tools/dbg/asan_tick.py "$@"
