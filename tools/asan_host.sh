#!/bin/bash
# tools/asan_host.sh [pytest args] — the library's HOST side under AddressSanitizer + UndefinedBehaviorSanitizer
# (VERDICT r5 #4): `make asan-host` builds 3dal_pytorch_amd/lib3dal_hip_asan.so (host code instrumented, device code the
# normal gfx950 code — GPU-side sanitizers are not available on the pool), and the CPU tests that call into the
# library (symbol table, argument checks incl. the 32-bit-extent limits, workspace sizing, pack sizing, the drop-in
# modules' host logic) run against it in this container: no GPU needed, every path they reach returns before a launch.
# ASan's runtime must be the first DSO of the (uninstrumented) python process: LD_PRELOAD. Leak checking is off
# (python itself "leaks" by design at exit); any ASan / UBSan report aborts the run (halt_on_error).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
make -C "$R/3dal_pytorch_amd/csrc" asan-host -j"${JOBS:-6}" >/dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$R"
export DAL3_TEST_LIB="$R/3dal_pytorch_amd/lib3dal_hip_asan.so"
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:verify_asan_link_order=0" UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
if [ $# -eq 0 ]; then set -- tests/test_host_cpu.py tests/test_host_eval.py tests/test_host_dropin_train.py; fi
LD_PRELOAD="$RT" python3 -m pytest "$@" -x -q -m "not gpu" -p no:cacheprovider
