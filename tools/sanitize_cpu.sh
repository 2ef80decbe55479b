#!/bin/bash
# tools/sanitize_cpu.sh — AddressSanitizer + UBSan over the CPU builds (GPU sanitizers are not available on this pool):
#   1. the oracle restatement (oracle/mlkem_oracle.c) under its golden / reference tests
#   2. the PRODUCT's kernel source compiled for the wave64 host emulator (tests/emu) under the emulated-kernel tests:
#      out-of-bounds LDS / global accesses and undefined behaviour in the kernels show up here.
# The instrumented libraries replace the normal ones for the duration of the run and are restored afterwards.
set -e
cd "$(dirname "$0")/.."
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -c "import __graft_entry__ as g; g.build()"
cp oracle/liboracle_mlkem.so /tmp/_oracle_plain.so
cp tests/emu/libmlkem_emu.so /tmp/_emu_plain.so
restore() { cp /tmp/_oracle_plain.so oracle/liboracle_mlkem.so; cp /tmp/_emu_plain.so tests/emu/libmlkem_emu.so; }
trap restore EXIT
if [ "${1:-all}" != emu ]; then   # `sanitize_cpu.sh emu` skips the oracle leg
gcc $SAN -fPIC -shared -std=c11 -o oracle/liboracle_mlkem.so oracle/mlkem_oracle.c
LD_PRELOAD="$PRE" python -m pytest tests/test_oracle_golden.py tests/test_oracle_vs_reference.py -x -q
cp /tmp/_oracle_plain.so oracle/liboracle_mlkem.so
fi
g++ $SAN -std=c++17 -pthread -fPIC -shared -Wno-unknown-pragmas -Wno-attributes -o tests/emu/libmlkem_emu.so tests/emu/emu_lib.cpp   # ~5 min
LD_PRELOAD="$PRE" python -m pytest tests/test_emulated_kernels.py tests/test_fips203_mode.py -x -q -m "not gpu"
