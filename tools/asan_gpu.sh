#!/bin/bash
# libtmc2hip.so with AddressSanitizer + UBSan on ALL host code (the .cpp files and the host side of the .hip files), built HERE
# into asan_build/ (travels to the GPU box with the snapshot; git-ignored), to run the benchmark / GPU tests against on the box:
#   tools/asan_gpu.sh                       # build (CPU, a few minutes)
#   gpurun -- 'bash tools/asan_gpu.sh run python bench.py --steps 20 --warmup 5 --cpu-baseline 0'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/asan_build
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so)
if [ "$1" = "run" ]; then
  shift
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
  cd "$ROOT"
  LD_PRELOAD=$RT TMC2_PACKAGE_DIR="$OUT/pkg" "$@"
  exit $?
fi
mkdir -p "$OUT/build" "$OUT/pkg"
cd "$ROOT/mpeg-pcc-tmc2_amd/csrc"
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -mavx2 -mbmi2 -mpopcnt -fsanitize=address,undefined -fno-sanitize=vptr -fno-gpu-sanitize -fno-omit-frame-pointer"
for f in *.cpp *.hip; do
  /opt/rocm/bin/hipcc $F -x hip -c "$f" -o "$OUT/build/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$OUT/pkg/libtmc2hip.so" "$OUT"/build/*.o
rm -rf "$OUT/pkg/tmc2_amd"; cp -r "$ROOT/mpeg-pcc-tmc2_amd/tmc2_amd" "$OUT/pkg/"
ls -la "$OUT/pkg/libtmc2hip.so"
