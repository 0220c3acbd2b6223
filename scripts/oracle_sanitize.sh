#!/bin/bash
# The CPU oracle (oracle/*.c) under AddressSanitizer + UndefinedBehaviorSanitizer: builds `make -C oracle sanitize`
# and runs the oracle's CPU tests (golden vectors, KATs, third-party pins, the fuzz harness' generators) against
# that build.  CPU only -- the GPU pool refuses sanitizer runs, and the HIP path is compared with this oracle.
# Exit status: pytest's (a sanitizer report aborts the process: -fno-sanitize-recover, ASAN halt_on_error).
set -euo pipefail
HERE="$(cd "$(dirname "$0")/.." && pwd)"
make -s -C "$HERE/oracle" sanitize
export SOSVO_ORACLE_LIB="$HERE/oracle/build/libsosvo_oracle_san.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
cd "$HERE"
python scripts/oracle_sanitize_flow.py
exec python -m pytest -q -x -m "not gpu" -p no:cacheprovider \
    tests/test_oracle_geometry.py tests/test_oracle_gp3p.py tests/test_oracle_image.py tests/test_oracle_match.py \
    tests/test_oracle_orb.py tests/test_oracle_ransac.py tests/test_oracle_relpose.py tests/test_oracle_thirdparty.py \
    tests/test_fuzz_harness_cpu.py "$@"
