set -x
export TMPDIR=/tmp
O=gpurun_out/r4c4; mkdir -p $O
for n in 8 4; do
  timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_$n.json 2> $O/tile_$n.err
  POMGPU_WR_NODEFER=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_nodefer.json 2>> $O/tile_$n.err
  POMGPU_WIDE_FULL=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_widefull.json 2>> $O/tile_$n.err
done
timeout -k 10 200 python tools/tile_probe.py --tiles 2 --rank 1 > $O/tile_2.json 2> $O/tile_2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c4/tile_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], d['tile'], 'wall', d['ms_per_step_wall'], 'ksum', d['kernel_ms_sum'], 'ext', k.get('k_ext_pair'), k.get('k_ext_ring'), 'rv', k.get('k_realvertvl_col'), 'prodl', k.get('k_profq_prod_lines'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?"; tail -c 300 $O/bench1.json
timeout -k 10 800 python -m pytest tests -m gpu -x -q --durations=8 > $O/gputests.log 2>&1; echo "gputests rc=$?"; tail -22 $O/gputests.log
