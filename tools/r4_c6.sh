set -x
export TMPDIR=/tmp
O=gpurun_out/r4c6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "short_wave or two_external or each_routine" > $O/gputests_sw.log 2>&1; echo "gputests rc=$?"; tail -5 $O/gputests_sw.log
POM_TILE_GRID=2x4 timeout -k 10 200 python tools/tile_probe.py --tiles 8 --rank 5 > $O/tile_8_2x4.json 2> $O/tile_8_2x4.err
POM_TILE_GRID=2x2 timeout -k 10 200 python tools/tile_probe.py --tiles 4 --rank 2 > $O/tile_4_2x2.json 2> $O/tile_4_2x2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c6/tile_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], d['tiles'], d['tile'], 'wall', d['ms_per_step_wall'], 'ksum', d['kernel_ms_sum'], 'ext', k.get('k_ext_pair'), k.get('k_ext_ring'), k.get('k_ext_step_adv'), 'profq', k.get('k_profq'))
    except Exception as e: print(f, 'ERR', e)
PY
bash tools/round_profile.sh r4 B
