set -x
export TMPDIR=/tmp
O=gpurun_out/r4c7; mkdir -p $O
for n in 8 4; do
  for rep in 1 2; do
    timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_base_$rep.json 2>> $O/tile_$n.err
    POMGPU_X_OVERLAP=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_ovl_$rep.json 2>> $O/tile_$n.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c7/tile_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], d['tile'], 'wall', d['ms_per_step_wall'], 'ksum', d['kernel_ms_sum'], 'ext', k.get('k_ext_pair'), 'profq', k.get('k_profq'), 'advt2', k.get('k_advt2x2_col'))
    except Exception as e: print(f, 'ERR', e)
PY
