set -x
export TMPDIR=/tmp
O=gpurun_out/r4c5; mkdir -p $O
for n in 8 4; do
  for rep in 1 2; do
    timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_last_$rep.json 2>> $O/tile_$n.err
    POMGPU_EXT_RING_FIRST=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_first_$rep.json 2>> $O/tile_$n.err
  done
  POMGPU_EXT_NOPAIR=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_nopair.json 2>> $O/tile_$n.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c5/tile_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], d['tile'], 'wall', d['ms_per_step_wall'], 'ksum', d['kernel_ms_sum'], 'ext', k.get('k_ext_pair'), k.get('k_ext_ring'), k.get('k_ext_step_adv'), 'profq', k.get('k_profq'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp32 or marching or two_external or call_sequence" > $O/gputests_rest.log 2>&1; echo "gputests rc=$?"; tail -12 $O/gputests_rest.log
timeout -k 10 600 python tools/swrad_drift.py > $O/swrad.log 2>&1; tail -12 $O/swrad.log
for w in seamount256 seamount65; do
  python bench.py --workload $w --steps 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w default', d['ms_per_step'], d['external_mode'])"
  POMGPU_EXT_PAIR=1 python bench.py --workload $w --steps 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w pair', d['ms_per_step'], d['external_mode'])"
done
