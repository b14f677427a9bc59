#!/bin/bash
# The checker under sanitizers: oracle/rtgr_oracle.cpp built with g++ -fsanitize=address,undefined and the oracle's CPU tests (the
# reference's goldens, the committed fixtures, the reference's unit tests, identities, true geodesics) run against that build.
# TEST INFRASTRUCTURE, like everything under oracle/.      usage: oracle/sanitize.sh [log]      (≈ 6 min; builds under oracle/_san/)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r05/sanitize_host.log}
mkdir -p "$ROOT/oracle/_san"
g++ -O1 -g -std=c++17 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -shared -o "$ROOT/oracle/_san/librtgr_oracle.so" "$ROOT/oracle/rtgr_oracle.cpp"
OTESTS="tests/test_oracle_golden.py tests/test_golden_fixtures.py tests/test_reference_unit_tests.py tests/test_identities.py tests/test_truth.py tests/test_user_objects.py"
echo "== asan + ubsan: oracle/rtgr_oracle.cpp (g++), $OTESTS -m 'not gpu'" | tee -a "$LOG"
(cd "$ROOT" && RTGR_ORACLE_LIB=$ROOT/oracle/_san/librtgr_oracle.so LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
   ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 python3 -m pytest $OTESTS -q -m "not gpu" 2>&1 | tail -n 4) | tee -a "$LOG"
rm -rf "$ROOT/oracle/_san"
