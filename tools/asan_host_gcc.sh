#!/bin/bash
# libtmc2hip.so with the HOST-ONLY translation units (C-ABI glue, packers, global patch allocation, orientation walk, k-d tree
# builder, PLY / checksum I/O, metric text) compiled by g++ with gcc's AddressSanitizer, linked with the device objects of the
# normal build (run `make -C mpeg-pcc-tmc2_amd/csrc` first).  gcc's ASan runtime does not intercept the HSA allocator (the
# ROCm clang runtime does, and died of it on the GPU box in round 3: gpurun_out/asan_bench.err), so this build runs next to
# the HIP runtime.  Built HERE into asan_build/ (travels with the snapshot; git-ignored):
#   tools/asan_host_gcc.sh                                   # build (CPU, a minute)
#   gpurun -- 'bash tools/asan_host_gcc.sh run python bench.py --steps 5 --warmup 2 --cpu-baseline 0'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/asan_build
RT=$(gcc -print-file-name=libasan.so)
if [ "$1" = "run" ]; then
  shift
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0:detect_odr_violation=0:verify_asan_link_order=0:use_sigaltstack=0
  cd "$ROOT"
  LD_PRELOAD="$RT $(gcc -print-file-name=libstdc++.so)" TMC2_PACKAGE_DIR="$OUT/pkg" "$@"   # (libstdc++ up front: the runtime resolves __cxa_throw when it starts)
  exit $?
fi
mkdir -p "$OUT/build" "$OUT/pkg"
cd "$ROOT/mpeg-pcc-tmc2_amd/csrc"
F="-std=c++17 -O1 -g -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../../include -I. -mavx2 -mbmi2 -mpopcnt -ffp-contract=off -fsanitize=address -fno-omit-frame-pointer"
for f in *.cpp; do
  g++ $F -c "$f" -o "$OUT/build/$f.o" &
done
wait
g++ -shared -fPIC -fsanitize=address -o "$OUT/pkg/libtmc2hip.so" "$OUT"/build/*.cpp.o build/*.hip.o -L/opt/rocm/lib -lamdhip64 -lpthread
rm -rf "$OUT/pkg/tmc2_amd"; cp -r "$ROOT/mpeg-pcc-tmc2_amd/tmc2_amd" "$OUT/pkg/"
ls -la "$OUT/pkg/libtmc2hip.so"
