"""developer tool (GPU box): time library variants / developer switches on the bench workload, one line per variant.
usage: python tools/sweep.py [--kernels advt2x2_col,advq2_col,...] [--workload W] spec ...   spec = tag[:lib=<path>][:ENV=VALUE]..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
kernels, wl = None, "basin2048"
while args and args[0].startswith("--"):
    if args[0] == "--kernels": kernels = args[1].split(","); args = args[2:]
    elif args[0] == "--workload": wl = args[1]; args = args[2:]
for spec in args:
    parts = spec.split(":")
    env = dict(os.environ)
    for p in parts[1:]:
        k, v = p.split("=", 1)
        if k == "lib": env["POMGPU_LIBPATH"] = os.path.join(ROOT, v)
        else: env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--profile-all", "--steps", "3", "--warmup", "2", "--workload", wl],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"{parts[0]:14s} FAILED: {r.stderr[-400:]}", flush=True); continue
    d = json.loads(line[-1])
    km = d["kernel_ms_per_step"]
    sel = {k[2:]: v for k, v in km.items() if kernels is None or k[2:] in kernels}
    print(f"{parts[0]:14s} step={d['ms_per_step']:.2f} int={d['internal_mode']['device_ms_per_step']:.2f} err={d['error_status']}  " + "  ".join(f"{k}={v}" for k, v in sel.items()), flush=True)
