set -x
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "digests or marching or two_external or general_kernels or fp32 or each_routine or call_sequence" > $O/t1.log 2>&1; tail -4 $O/t1.log
for n in 8 4; do
  timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_ahead.json 2> $O/tile_$n.err
  POMGPU_EXT_NOAHEAD=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_noahead.json 2>> $O/tile_$n.err
  for r in 6 12 16; do POMGPU_EXT_AHEAD_ROWS=$r timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_ahead_r$r.json 2>> $O/tile_$n.err; done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/tile_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['tile'], 'wall', d['ms_per_step_wall'], 'ext', d['kernels'].get('k_ext_step_adv'))
    except Exception as e: print(f, 'ERR', e)
PY
for w in seamount256 seamount65; do
  python bench.py --workload $w --steps 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w ahead', d['ms_per_step'], d['external_mode'])"
  POMGPU_EXT_NOAHEAD=1 python bench.py --workload $w --steps 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w noahead', d['ms_per_step'], d['external_mode'])"
done
timeout -k 10 600 python tests/gpu_tiles_threads.py 2048x1536x50 4 50 > $O/threads_full.log 2>&1; tail -3 $O/threads_full.log
timeout -k 10 300 python tests/gpu_tiles_threads.py 256x192x50 4 6 f32 > $O/threads_f32.log 2>&1; tail -3 $O/threads_f32.log
timeout -k 10 600 python tools/swrad_drift.py > $O/swrad.log 2>&1; tail -12 $O/swrad.log
