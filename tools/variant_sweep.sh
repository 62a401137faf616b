#!/bin/bash
# developer tool (GPU box): time the library variants of build_variants/ on the bench workload
O=gpurun_out/${1:-variants}; mkdir -p $O
python tools/bench_brief.py --steps 3 --warmup 2 > $O/base.txt 2>&1; echo "base: $(head -c 300 $O/base.txt | head -1)"; grep -o "profq=[0-9.]*" $O/base.txt | head -1
for f in build_variants/libpomgpu_*.so; do
  t=$(basename $f .so | sed 's/libpomgpu_//')
  POMGPU_LIBPATH=$PWD/$f python tools/bench_brief.py --steps 3 --warmup 2 > $O/$t.txt 2>&1
  echo "$t: $(head -1 $O/$t.txt | cut -c1-60) $(grep -o 'profq=[0-9.]*' $O/$t.txt | head -1)"
done
