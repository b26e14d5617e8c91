#!/bin/bash
# The oracle (oracle/orl_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU: every golden fixture is
# replayed through the instrumented build.  (GPU sanitizers are not available on the pool; the HIP side is covered by the
# parity tests.)  usage: tools/sanitize_oracle.sh [out.txt]
set -e
cd "$(dirname "$0")/.."
out=${1:-profiles/sanitizer_oracle.txt}
make -C oracle -s asan
export ORL_ORACLE_SO=$PWD/oracle/_build/liborloracle_asan.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{ echo "# $(gcc --version | head -1); -fsanitize=address,undefined; $(date -u +%F)";
  python -m pytest tests/test_oracle_golden.py tests/test_sharding.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -5; } | tee "$out"
