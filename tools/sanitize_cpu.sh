#!/usr/bin/env bash
# tools/sanitize_cpu.sh — AddressSanitizer + UBSan over the CPU-side code (the oracle restatement and the host layer's
# Matrix Market reader / vector files), driven by the CPU tests.  GPU sanitizers are not available on the pool; the
# device code is covered by host-side shape checks before every launch and by the parity tests.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/spmv_asan.XXXXXX)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
gcc -std=c11 $SAN -fPIC -fopenmp -ffp-contract=off -c "$R/oracle/spmv_oracle.c" -o "$T/plain.o"
gcc -std=c11 $SAN -fPIC -mfma -ffp-contract=off -DORC_FMA -c "$R/oracle/spmv_oracle.c" -o "$T/fma.o"
gcc -shared -fopenmp $SAN -o "$T/libspmv_oracle.so" "$T/plain.o" "$T/fma.o" -lm
g++ -std=c++17 $SAN -fPIC -pthread -I"$R/include" -I"$R/arm-spmv_amd/host" -shared "$R/arm-spmv_amd/host/compat.cpp" \
    "$R/arm-spmv_amd/host/mtx_io.cpp" -o "$T/libarmspmv_compat.so" -L"$R/arm-spmv_amd/lib" -lspmv_hip -Wl,-rpath,"$R/arm-spmv_amd/lib"
cd "$R"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 SPMV_ORACLE_SO="$T/libspmv_oracle.so" SPMV_COMPAT_SO="$T/libarmspmv_compat.so" \
    python3 -m pytest tests/test_oracle_golden.py tests/test_oracle_symgs.py tests/test_host_io.py -x -q 2>&1 | grep -E "passed|failed|ERROR|runtime error|Sanitizer"
rm -rf "$T"
