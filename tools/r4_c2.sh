set -x
export TMPDIR=/tmp
O=gpurun_out/r4c2; mkdir -p $O
python - <<'PY'
import os, subprocess, sys
for v in ("plain", "hidden"):
    ps = []
    for r in (0, 1):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617" if v == "plain" else "29618")
        if v == "hidden":
            env.update(HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        ps.append(subprocess.Popen([sys.executable, "tools/diag/kfd_open.py"], env=env, stdout=open(f"gpurun_out/r4c2/kfd_{v}_{r}.txt", "w"), stderr=subprocess.STDOUT))
    for p in ps:
        p.wait(timeout=300)
PY
cat $O/kfd_plain_0.txt $O/kfd_hidden_0.txt
for n in 8 4; do
  POMGPU_EXT_NOAHEAD=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_noahead.json 2>> $O/tile_$n.err
  for r in 5 8 12 16 24; do POMGPU_EXT_AHEAD_ROWS=$r timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_ahead_r$r.json 2>> $O/tile_$n.err; done
  POMGPU_EXT_PAIR=1 timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_${n}_pair.json 2>> $O/tile_$n.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c2/tile_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], d['tile'], 'wall', d['ms_per_step_wall'], 'ext', k.get('k_ext_step_adv'), k.get('k_ext_pair'), k.get('k_ext_ring'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?"; tail -c 300 $O/bench1.json
