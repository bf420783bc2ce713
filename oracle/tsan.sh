#!/bin/bash
# ThreadSanitizer + Archer (OpenMP-aware) run of the oracle's OpenMP loops: oracle/tsan_check.cpp.
# Needs clang with libomp / libarcher (the ROCm LLVM has them); writes the report to stdout.
#   oracle/tsan.sh > profiles/r03_oracle_tsan.txt 2>&1
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
LLVM=${LLVM:-/opt/rocm/lib/llvm}
OUT=${TMPDIR:-/tmp}/gpd_oracle_tsan
$LLVM/bin/clang++ -O1 -g -fopenmp -fsanitize=thread -ffp-contract=off -mfma -mavx2 -std=c++17 -Wno-unused-function \
  -o $OUT $HERE/tsan_check.cpp $HERE/gpd_oracle.cpp -Wl,-rpath,$LLVM/lib || { echo "tsan.sh: build failed"; exit 2; }
echo "# $(date -u +%F) clang $($LLVM/bin/clang++ --version | head -1)"
echo "# OMP_NUM_THREADS=8 OMP_TOOL_LIBRARIES=libarcher.so TSAN_OPTIONS=ignore_noninstrumented_modules=1"
OMP_NUM_THREADS=8 OMP_TOOL_LIBRARIES=$LLVM/lib/libarcher.so ARCHER_OPTIONS="verbose=1" \
  TSAN_OPTIONS="ignore_noninstrumented_modules=1 halt_on_error=0" $OUT
rc=$?
echo "# positive control (a deliberately racy OpenMP loop, same build): ThreadSanitizer warnings counted below"
OMP_NUM_THREADS=8 OMP_TOOL_LIBRARIES=$LLVM/lib/libarcher.so TSAN_OPTIONS="ignore_noninstrumented_modules=1 halt_on_error=0" \
  $OUT --positive-control 2>&1 | grep -c "WARNING: ThreadSanitizer: data race" | sed 's/^/# data-race reports in the positive control: /'
echo "# exit status $rc (0 = no race reported, results identical; 66 = ThreadSanitizer reported races)"
exit $rc
