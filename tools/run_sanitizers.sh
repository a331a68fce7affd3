#!/bin/bash
# Sanitizer pass over the CPU-side code: no GPU involved.
#   1. oracle/eccx_oracle.c under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the
#      oracle's own CPU tests (golden vectors, RFC 7748, gloo sharding through the oracle);
#   2. the C ABI's host code (eccoxide_amd/csrc/eccx_api.cpp, host side only: -fno-gpu-sanitize)
#      under the same sanitizers, driven by the argument / error-path tests that need no device.
# usage: bash tools/run_sanitizers.sh [outfile]     (writes a short report, default profiles/r03_sanitizers.txt)
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/profiles/r03_sanitizers.txt}
cd "$ROOT"
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
{
echo "# sanitizer pass, $(date -u +%Y-%m-%dT%H:%MZ), $(gcc --version | head -1)"
echo "## 1. oracle (gcc -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined)"
make -C oracle SAN=1 -s || exit 1
LD_PRELOAD="$ASAN:$UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  ECCX_ORACLE_SAN=1 python -m pytest tests/test_oracle_golden.py tests/test_x25519.py tests/test_dist_gloo.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
echo "## 2. C ABI host code (hipcc -fsanitize=address,undefined -fno-gpu-sanitize, host objects only)"
make -C eccoxide_amd/csrc san -s || exit 1
CLANG_RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD="$CLANG_RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  ECCX_LIB_PATH=$ROOT/eccoxide_amd/libeccx_san.so python -m pytest tests/test_abi.py tests/test_host_logic.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
} | tee "$OUT"
