#!/bin/bash
# tools/oracle_sanitize.sh: the CPU oracle (oracle/oracle.c, test infrastructure) built with AddressSanitizer + UBSan and run through
# its CPU tests (no GPU needed; sanitizers are for the CPU build only on this pool).  The regular build is put back afterwards.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
make -s -C "$root/oracle"
cp "$root/oracle/_build/liboracle.so" "$tmp/orig.so"
trap 'cp "$tmp/orig.so" "$root/oracle/_build/liboracle.so"; rm -rf "$tmp"' EXIT
gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -fno-fast-math -D_POSIX_C_SOURCE=200809L -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o "$root/oracle/_build/liboracle.so" "$root/oracle/oracle.c" -lm -lpthread
cd "$root"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    python -m pytest tests/test_oracle_cpu.py tests/test_camera_cpu.py -x -q
