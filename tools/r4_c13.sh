set -x
export TMPDIR=/tmp
O=gpurun_out/r4c13; mkdir -p $O
export POMGPU_LIBPATH=$PWD/build_variants/libpomgpu_xovl.so
for n in 8 4; do
  for rep in 1 2; do
    timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_base_$rep.json 2>> $O/tile_$n.err
    POMGPU_X_OVERLAP=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_ovl_$rep.json 2>> $O/tile_$n.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c13/tile_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], d['tile'], 'wall', d['ms_per_step_wall'], 'ksum', d['kernel_ms_sum'], 'profq', k.get('k_profq'), 'advt2', k.get('k_advt2x2_col'))
    except Exception as e: print(f, 'ERR', e)
PY
POMGPU_X_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/tile_probe.py --tiles 8 --rank 4 --steps 3 > $O/probe.json 2> $O/probe.err
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r4c13/trace/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [n for n, r in enumerate(rows) if 'k_profq<' in r['Kernel_Name']]
n0 = idx[-1]
t0 = int(rows[n0 - 6]['Start_Timestamp'])
for r in rows[n0 - 6:n0 + 8]:
    print(f"{r['Kernel_Name'][:40]:40s} queue {r.get('Queue_Id')} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us end {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us")
PY
