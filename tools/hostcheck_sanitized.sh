#!/bin/bash
# The kernels' arithmetic headers (filterpy_amd/csrc/fk_math*.hpp, fk_ukf*.hpp, fk_imm.hpp, fk_exact_scan.hpp, fk_resample_*.hpp) as
# the host harness compiles them (tests/hostcheck/), under AddressSanitizer + UndefinedBehaviorSanitizer: every hostcheck test once
# more with out-of-bounds indices, signed overflow, bad shifts and misaligned accesses fatal.  CPU only; ~4 minutes.
#     bash tools/hostcheck_sanitized.sh [summary-file]
# The sanitized libraries replace tests/hostcheck/lib*.so for the run and are removed afterwards (conftest rebuilds the plain ones).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
HC=$R/tests/hostcheck
OUT=${1:-/dev/stdout}
SAN="-fsanitize=address,undefined -fno-sanitize-recover=all -g -fno-omit-frame-pointer"
ASAN=$(gcc -print-file-name=libasan.so)
cd "$HC" || exit 2
g++ -O1 -std=c++17 -fPIC -shared -ffp-contract=on -w $SAN -o libhostcheck.so hostcheck.cpp &&
g++ -O1 -std=c++17 -fPIC -shared -ffp-contract=on -w $SAN -o libhostcheck_quad.so hostcheck_quad.cpp &&
g++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off -w $SAN -o libhostcheck_rs.so hostcheck_rs.cpp || { rm -f lib*.so; exit 2; }
cd "$R"
LOG=$(mktemp)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests/test_hostcheck_math.py tests/test_hostcheck_imm.py tests/test_hostcheck_exact_scan.py \
    tests/test_hostcheck_resample_math.py tests/test_hostcheck_ukf.py tests/test_hostcheck_ukf_quad.py -q -p no:cacheprovider > "$LOG" 2>&1
rc=$?
rm -f "$HC"/lib*.so                      # the plain ones are rebuilt by tests/conftest.py on the next run
{
    echo "hostcheck under -fsanitize=address,undefined (gcc $(gcc -dumpversion)), $(date -u +%F)"
    tail -1 "$LOG"
    echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' "$LOG")"
} > "$OUT"
[ $rc -ne 0 ] && tail -40 "$LOG"
rm -f "$LOG"
exit $rc
