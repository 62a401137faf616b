set -x
export TMPDIR=/tmp
O=gpurun_out/r4c3; mkdir -p $O
python - <<'PY'
import os, subprocess, sys
for v, be in (("gloo_nobarrier", "gloo"), ("cpugloo", "cpu:gloo")):
    ps = []
    for r in (0, 1):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617" if v == "cpugloo" else "29618", DIAG_BACKEND=be,
                   DIAG_BARRIER="1" if v == "cpugloo" else "0")
        ps.append(subprocess.Popen([sys.executable, "tools/diag/kfd_open.py"], env=env, stdout=open(f"gpurun_out/r4c3/kfd_{v}_{r}.txt", "w"), stderr=subprocess.STDOUT))
    for p in ps:
        p.wait(timeout=300)
PY
grep -h "^\[0\]" $O/kfd_gloo_nobarrier_0.txt $O/kfd_cpugloo_0.txt
for n in 8 4; do timeout -k 10 200 python tools/tile_probe.py --tiles $n --rank $((n/2)) > $O/tile_$n.json 2> $O/tile_$n.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c3/tile_*.json')):
    try:
        d=json.load(open(f)); k=d['kernels']; print(f.split('/')[-1], d['tile'], 'wall', d['ms_per_step_wall'], 'ext', k.get('k_ext_step_adv'), k.get('k_ext_pair'), k.get('k_ext_ring'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 950 python -m pytest tests -m gpu -x -q --durations=15 > $O/gputests.log 2>&1; echo "gputests rc=$?"; tail -30 $O/gputests.log
