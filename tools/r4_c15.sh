set -x
export TMPDIR=/tmp
O=gpurun_out/r4c15; mkdir -p $O
V=$PWD/build_variants/libpomgpu_edge8.so
for rep in 1 2; do
  timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 4 > $O/tile_8_base_$rep.json 2>> $O/tile_8.err
  POMGPU_LIBPATH=$V timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 4 > $O/tile_8_edge8_$rep.json 2>> $O/tile_8.err
done
python - <<'PY'
import json,glob
ks=('k_bcond6_edges','k_bcond4_edges','k_bcondorl3','k_q_filter_rim','k_profq_rim','k_profq_prod_lines','k_halo_pack8','k_halo_unpack8')
for f in sorted(glob.glob('gpurun_out/r4c15/tile_*.json')):
    d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], 'wall', d['ms_per_step_wall'], {x: k.get(x,[0,0])[1] for x in ks})
PY
for w in basin2048 seamount256; do
  python bench.py --workload $w --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$w base', d['ms_per_step'], {x:k.get(x) for x in ('k_bcond6_edges','k_bcond4_edges','k_bcondorl3','k_q_filter_rim','k_profq_rim')})"
  POMGPU_LIBPATH=$V python bench.py --workload $w --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$w edge8', d['ms_per_step'], {x:k.get(x) for x in ('k_bcond6_edges','k_bcond4_edges','k_bcondorl3','k_q_filter_rim','k_profq_rim')})"
done
