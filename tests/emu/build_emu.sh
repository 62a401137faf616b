#!/bin/bash
# TEST INFRASTRUCTURE: host build of the unmodified kernel sources (see tests/emu/hip/hip_runtime.h)
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
OUT=$ROOT/tests/_emu; mkdir -p "$OUT"
SRC=$ROOT/extpom_amd/csrc
pids=()
FLAGS="-x c++ -std=c++17 -O2 -ffp-contract=off -fno-fast-math -fPIC -w -I$HERE -I$ROOT/include -I$SRC"
for f in k_ext k_adv k_vert k_tile k_bc pomgpu_api transport cdf_out; do
  g++ $FLAGS -c "$SRC/$f.hip" -o "$OUT/$f.o" & pids+=($!)
done
g++ $FLAGS -c "$HERE/emu_support.cpp" -o "$OUT/emu_support.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done     # a failed compile fails the build (plain `wait` would hide it)
g++ -shared -o "$OUT/libpomgpu_emu.so" "$OUT"/*.o -lm
echo "built $OUT/libpomgpu_emu.so"
