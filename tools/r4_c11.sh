set -x
export TMPDIR=/tmp
O=gpurun_out/r4c11; mkdir -p $O
for n in 8 4; do
  timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_base.json 2>> $O/tile_$n.err
  POMGPU_LIBPATH=$PWD/build_variants/libpomgpu_lds4.so timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_lds4.json 2>> $O/tile_$n.err
  POMGPU_LIBPATH=$PWD/build_variants/libpomgpu_lds6.so timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_lds6.json 2>> $O/tile_$n.err
  POMGPU_PROFQ_ROWS2=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_profq2.json 2>> $O/tile_$n.err
  timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_base2.json 2>> $O/tile_$n.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c11/tile_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], d['tile'], 'wall', d['ms_per_step_wall'], {x: k.get(x, [0,0])[1] for x in ('k_profq','k_advt2x2_col','k_advq2_col','k_advuv_col','k_advct_col','k_ts_update')})
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "never_open" 2>&1 | tail -3
