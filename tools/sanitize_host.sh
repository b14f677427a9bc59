#!/bin/bash
# Host-side sanitizer runs of the library (no GPU needed; GPU ASan is not available on this pool): librtgr_hip.so rebuilt with
# AddressSanitizer, then with UndefinedBehaviorSanitizer, on the HOST code only (-fno-gpu-sanitize), and the CPU tests that call into
# the library run against each — the ISA audit on fuzzed code objects, the listing repair, the in-process unit build (hiprtc +
# comgr), rtgr_user_source_join, the argument checks of every entry point.
# (The checker has the same run of its own: oracle/sanitize.sh.)
#     usage: tools/sanitize_host.sh [log]          (≈ 10 min on 8 cores; builds under raytracegr.jl_amd/build/{asan,ubsan}/)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r05/sanitize_host.log}
RTDIR=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | tail -n 1)
TESTS="tests/test_abi.py tests/test_build_checks.py tests/test_host_logic.py tests/test_user_objects.py tests/test_unit_probe.py tests/test_user_metric.py"
: > "$LOG"
for kind in asan ubsan; do
  if [ $kind = asan ]; then FLAGS='"-fsanitize=address"'; RT=$RTDIR/libclang_rt.asan-x86_64.so
  else FLAGS='"-fsanitize=undefined", "-fno-sanitize=vptr,function", "-fno-sanitize-recover=undefined"'; RT=$RTDIR/libclang_rt.ubsan_standalone-x86_64.so; fi
  mkdir -p "$ROOT/raytracegr.jl_amd/build/$kind"
  python3 - <<PY
import importlib.util
spec = importlib.util.spec_from_file_location("b", "$ROOT/raytracegr.jl_amd/build.py")
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.build(extra=[$FLAGS, "-shared-libsan", "-fno-gpu-sanitize", "-g", "-fno-omit-frame-pointer", "-O1"], verbose=False,
        out="$ROOT/raytracegr.jl_amd/build/$kind/librtgr_hip.so", obj_dir="$ROOT/raytracegr.jl_amd/build/$kind/obj")
PY
  echo "== $kind: host code of librtgr_hip.so, $TESTS -m 'not gpu'" | tee -a "$LOG"
  (cd "$ROOT" && RTGR_CSRC=$ROOT/raytracegr.jl_amd/csrc RTGR_LIB=$ROOT/raytracegr.jl_amd/build/$kind/librtgr_hip.so LD_PRELOAD=$RT \
     ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 python3 -m pytest $TESTS -q -m "not gpu" 2>&1 | tail -n 4) | tee -a "$LOG"
done
rm -rf "$ROOT/raytracegr.jl_amd/build/asan" "$ROOT/raytracegr.jl_amd/build/ubsan"   # (≈ 70 MB that would otherwise travel to the GPU box with every gpurun call)
