"""developer tool: run bench.py and print a compact per-kernel table (ms per step)"""
import json, subprocess, sys
args = sys.argv[1:]
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--profile-all"] + args, capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-3000:]); sys.exit(1)
d = json.loads(line[-1])
print(f"{d['ms_per_step']:.2f} ms/step  value={d['value']:.4g}  internal={d['internal_mode']['device_ms_per_step']} ms  external={d['external_mode']['device_ms_per_step']} ms  roofline={d['roofline']}")
print("  ".join(f"{k[2:]}={v}" for k, v in d["kernel_ms_per_step"].items()))
