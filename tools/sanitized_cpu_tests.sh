#!/bin/bash
# The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU pool allows no sanitizer runs):
# the oracle (oracle/libsmpc_oracle*.so) and the C++ host adapter (host/libsmpc_host.so, host_cpu_tests) are rebuilt
# with -fsanitize=address,undefined, the whole CPU test suite runs against them with the sanitizer runtime preloaded
# into the Python process, then the normal builds are restored. Any report fails the run (halt_on_error, no recovery).
# usage: tools/sanitized_cpu_tests.sh [pytest args]        (from the repository root)
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
make -C oracle -s asan && make -C nav2_social_mpc_controller_amd/host -s asan || { echo "sanitized build failed"; exit 2; }
# tests/native builds its own shim with the default flags; the oracle's counting library is sanitized with the rest
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=99"   # leaks: CPython's and HIP's own at exit
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
PRE="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
LD_PRELOAD="$PRE" ./nav2_social_mpc_controller_amd/host/host_cpu_tests; rc_host=$?
LD_PRELOAD="$PRE" python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"; rc=$?
make -C oracle -s clean && make -C oracle -s && make -C nav2_social_mpc_controller_amd/host -s clean && make -C nav2_social_mpc_controller_amd/host -s
echo "sanitized CPU suite: host_cpu_tests rc=$rc_host, pytest rc=$rc"
[ "$rc" = 0 ] && [ "$rc_host" = 0 ]
