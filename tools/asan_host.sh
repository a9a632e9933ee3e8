#!/bin/bash
# The host translation units of libtmc2hip.so (packers, global patch allocation, PLY / checksum I/O, k-d tree builder,
# orientation walk, C-ABI glue) rebuilt with AddressSanitizer + UBSan and the CPU test tier / fuzzers run against that
# build.  Device objects are taken from the normal build (run `make -C mpeg-pcc-tmc2_amd/csrc` first).  Everything goes
# to a scratch directory; the in-tree library is not touched.
#   tools/asan_host.sh [scratch-dir]            (round 1: host tests, fuzz_gpa 0-500, fuzz_ply 0-300, fuzz_seg2 0-60: clean)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/tmc2_asan}
mkdir -p "$OUT/build" "$OUT/pkg"
cd "$ROOT/mpeg-pcc-tmc2_amd/csrc"
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -mavx2 -mbmi2 -mpopcnt"
for f in *.cpp; do
  /opt/rocm/bin/hipcc $F -fsanitize=address,undefined -fno-sanitize=vptr -fno-gpu-sanitize -fno-omit-frame-pointer -x hip -c "$f" -o "$OUT/build/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$OUT/pkg/libtmc2hip.so" "$OUT"/build/*.o build/*.hip.o
cp -r "$ROOT/mpeg-pcc-tmc2_amd/tmc2_amd" "$OUT/pkg/"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
cd "$ROOT"
LD_PRELOAD=$RT TMC2_PACKAGE_DIR="$OUT/pkg" python -m pytest tests/test_host_logic.py -x -q -m "not gpu" -p no:cacheprovider
cp tools/fuzz/fuzz_seg.py "$OUT/"   # (fuzz_seg2.py reads its cloud generators from it)
for s in "fuzz_gpa.py 0 100" "fuzz_ply.py 0 100" "fuzz_seg2.py 0 20"; do
  set -- $s
  sed "s#os.path.join(R,'mpeg-pcc-tmc2_amd')#'$OUT/pkg'#" tools/fuzz/$1 | sed "s#^import os; R=.*abspath(__file__)))); #import os; R='$ROOT'; #" > "$OUT/$1"
  LD_PRELOAD=$RT python "$OUT/$1" $2 $3 | tail -1
done
