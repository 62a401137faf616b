#!/bin/bash
# developer tool: libpomgpu.so with extra -D flags -> build_variants/libpomgpu_<tag>.so (load with POMGPU_LIBPATH)
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); SRC=$ROOT/extpom_amd/csrc
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fPIC -shared "$@" -I$ROOT/include -I$SRC \
  -o $ROOT/build_variants/libpomgpu_$TAG.so $SRC/k_ext.hip $SRC/k_adv.hip $SRC/k_vert.hip $SRC/k_tile.hip $SRC/k_bc.hip $SRC/k_reduce.hip $SRC/pomgpu_api.hip $SRC/transport.hip $SRC/cdf_out.hip
echo built $TAG
