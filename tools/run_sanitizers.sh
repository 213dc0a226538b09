#!/bin/bash
# CPU ONLY (build container): AddressSanitizer + UndefinedBehaviorSanitizer over the host-side native code --
#   phylo_hmrf_amd/csrc/ou_host.cpp, preprocess_host.cpp  (libphmrf_host.so: M-step objective, SLSQP driver, K-state threads,
#                                                          median fill, bilateral filter)
#   oracle/estep_oracle.c                                  (the oracle's C restatement)
# The tests that drive them run against the instrumented builds (PHMRF_HOST_LIB / PHMRF_ORACLE_LIB), python under
# LD_PRELOAD=libasan.  The GPU pool refuses sanitizer runs; nothing here touches a GPU.   -> profiles/r4_sanitizers.log
set -e
cd "$(dirname "$0")/.."
make -s -C phylo_hmrf_amd/csrc asan
make -s -C oracle asan
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
export LD_PRELOAD="$ASAN:$UBSAN"
# detect_leaks=0: CPython itself leaks at exit.  alloc_dealloc_mismatch=0: two of the tests load the REFERENCE's gco
# (oracle/_ref/libgco_ref.so, uninstrumented, the checker) whose max-flow frees a new[] array with delete
# (Graph<>::maxflow; ASan's interceptors see it in any library) -- not this repository's code and not ours to change.
export ASAN_OPTIONS="detect_leaks=0:alloc_dealloc_mismatch=0:abort_on_error=0:halt_on_error=1:exitcode=66"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=67"
export PHMRF_HOST_LIB="$PWD/phylo_hmrf_amd/libphmrf_host_asan.so"
export PHMRF_ORACLE_LIB="$PWD/oracle/libphmrf_oracle_asan.so"
LOG=profiles/r4_sanitizers.log
{
  echo "# tools/run_sanitizers.sh: $(gcc --version | head -1); -fsanitize=address,undefined -O1 -g"
  echo "# PHMRF_HOST_LIB=$PHMRF_HOST_LIB"
  echo "# PHMRF_ORACLE_LIB=$PHMRF_ORACLE_LIB"
  echo "# ASAN_OPTIONS=$ASAN_OPTIONS"
  set +e
  python -m pytest tests/test_mstep.py tests/test_preprocess.py tests/test_oracle_c.py tests/test_host_logic.py tests/test_em_driver.py -q -p no:cacheprovider 2>&1
  echo "# pytest exit code $?"
  set -e
} | tee $LOG
if grep -q "ERROR: AddressSanitizer\|runtime error:" $LOG || ! grep -q "# pytest exit code 0" $LOG; then echo "SANITIZER FINDINGS (or a failing test)"; exit 1; fi
echo "sanitizers clean" | tee -a $LOG
