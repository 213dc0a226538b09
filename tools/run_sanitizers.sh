#!/bin/bash
# CPU ONLY (build container): AddressSanitizer + UndefinedBehaviorSanitizer over the host-side native code --
#   phylo_hmrf_amd/csrc/ou_host.cpp, preprocess_host.cpp  (libphmrf_host.so: M-step objective, SLSQP driver, K-state threads,
#                                                          median fill, bilateral filter)
#   oracle/estep_oracle.c                                  (the oracle's C restatement)
# The tests that drive them run against the instrumented builds (PHMRF_HOST_LIB / PHMRF_ORACLE_LIB), python under
# LD_PRELOAD=libasan.  The GPU pool refuses sanitizer runs; nothing here touches a GPU.   -> profiles/r5_sanitizers.log
set -e
cd "$(dirname "$0")/.."
make -s -C phylo_hmrf_amd/csrc asan
make -s -C oracle asan
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
export LD_PRELOAD="$ASAN:$UBSAN"
# Two passes.  (1) The tests that never load the reference's gco, with EVERY check on.  (2) The tests that do load it
# (oracle/_ref/libgco_ref.so, uninstrumented, the checker): its max-flow frees a new[] array with delete (Graph<>::maxflow;
# ASan's interceptors see that in any library) -- not this repository's code and not ours to change --, so that pass alone
# runs with alloc_dealloc_mismatch=0.  detect_leaks=0 in both: CPython itself leaks at exit.
COMMON="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=66"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=67"
export PHMRF_HOST_LIB="$PWD/phylo_hmrf_amd/libphmrf_host_asan.so"
export PHMRF_ORACLE_LIB="$PWD/oracle/libphmrf_oracle_asan.so"
export ASAN_OPTIONS="$COMMON"
LOG=profiles/r5_sanitizers.log
WITH_GCO="tests/test_oracle_c.py tests/test_em_driver.py"
WITHOUT_GCO="tests/test_mstep.py tests/test_preprocess.py tests/test_host_logic.py tests/test_tiles_cpu.py"
{
  echo "# tools/run_sanitizers.sh: $(gcc --version | head -1); -fsanitize=address,undefined -O1 -g"
  echo "# PHMRF_HOST_LIB=$PHMRF_HOST_LIB"
  echo "# PHMRF_ORACLE_LIB=$PHMRF_ORACLE_LIB"
  set +e
  export ASAN_OPTIONS="$COMMON"
  echo "# pass 1 (no gco loaded, every check on): ASAN_OPTIONS=$ASAN_OPTIONS"
  python -m pytest $WITHOUT_GCO -q -p no:cacheprovider 2>&1
  echo "# pass 1 pytest exit code $?"
  export ASAN_OPTIONS="$COMMON:alloc_dealloc_mismatch=0"
  echo "# pass 2 (the tests that load the reference's gco): ASAN_OPTIONS=$ASAN_OPTIONS"
  python -m pytest $WITH_GCO -q -p no:cacheprovider 2>&1
  echo "# pass 2 pytest exit code $?"
  set -e
} | tee $LOG
unset LD_PRELOAD
if grep -q "ERROR: AddressSanitizer\|runtime error:" $LOG || ! grep -q "# pass 1 pytest exit code 0" $LOG || ! grep -q "# pass 2 pytest exit code 0" $LOG; then echo "SANITIZER FINDINGS (or a failing test)"; exit 1; fi
echo "sanitizers clean" | tee -a $LOG
