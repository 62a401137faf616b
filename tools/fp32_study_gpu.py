"""BASELINE configs[4] -- "fp32 internal mode / fp64 external mode: tolerance + GB/s study" -- on the GPU.
Two builds of the SAME sources run side by side: libpomgpu.so (every array fp64: the product) and libpomgpu_f32.so
(-DPOMGPU_STORE_F32: the 3-D arrays stored as fp32, every value widened to fp64 on load and rounded on store; all
arithmetic, the Thomas solves, the vertical integrals and the whole 2-D external mode fp64).
  part A (tolerance): seamount 65x49x21 and basin 256x192x50 stepped by both; after 1, 10, 100, 1000 internal steps the
                      largest difference of every prognostic field, relative to the field's largest magnitude
  part B (GB/s):      2048x1536x50, the two builds interleaved in one process (tools/kbench.py's method): ms per kernel,
                      internal mode, algorithmic GB/s at 8 and at 4 bytes per 3-D value
  part C (tolerance on the config's OWN grid): 2048x1536x50, both builds in one process (60 + 30 GB of the 288), after 1, 10, 100
                      internal steps the largest difference of every prognostic field, relative to the field's largest magnitude
                      (--full-drift [STEPS,...]; written to --drift-out, default profiles/round5_fp32_drift_basin2048.json)
usage: python tools/fp32_study_gpu.py [--skip-speed] [--skip-small] [--full-drift [1,10,100]] [--out gpurun_out/fp32_study.json]"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from extpom_amd import dist as pdist, lib as L
from extpom_amd.cases import make_case
from extpom_amd.layout import PROGNOSTIC
from extpom_amd.model import PomGpu, gpu_finish_initial

out_path = "gpurun_out/fp32_study.json"
if "--out" in sys.argv:
    out_path = sys.argv[sys.argv.index("--out") + 1]
res = {"builds": {}, "tolerance": {}, "speed": {}}


def rel(a, b, f):
    x, y = a.field(f), b.field(f)
    return float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-300))


from fp32_drift import full_size_drift   # tools/fp32_drift.py: also used by tests/test_gpu_parity.py


if "--full-drift" in sys.argv:
    k = sys.argv.index("--full-drift")
    steps_list = [int(v) for v in sys.argv[k + 1].split(",")] if k + 1 < len(sys.argv) and sys.argv[k + 1][0].isdigit() else [1, 10, 100]
    drift_out = sys.argv[sys.argv.index("--drift-out") + 1] if "--drift-out" in sys.argv else os.path.join(ROOT, "profiles", "round5_fp32_drift_basin2048.json")
    d = full_size_drift(steps_list)
    os.makedirs(os.path.dirname(drift_out) or ".", exist_ok=True)
    json.dump(d, open(drift_out, "w"), indent=1)
    print("wrote", drift_out)
    res["tolerance_full_size"] = d

for case, dims in (() if "--skip-small" in sys.argv else (("seamount", (65, 49, 21)), ("basin", (256, 192, 50)))):
    a = make_case(case, *dims, dte=6.0, isplit=30)
    gpu_finish_initial(a, device=0)
    b = a.copy()
    g64, g32 = PomGpu(a, device=0), PomGpu(b, device=0, libpath=L.LIBPATH_F32)
    res["builds"] = {"fp64": g64.L.pomgpu_version().decode(), "fp32-storage": g32.L.pomgpu_version().decode()}
    done, rows = 0, {}
    for n in (1, 10, 100, 1000) if dims[0] == 65 else (1, 10, 100):
        g64.run(n - done); g32.run(n - done); done = n
        g64.download(); g32.download()
        rows[str(n)] = {f: rel(a, b, f) for f in PROGNOSTIC}
        rows[str(n)]["error_status"] = [int(a.error_status), int(b.error_status)]
        print(f"{case} {dims} step {n:5d}: " + "  ".join(f"{f}={rows[str(n)][f]:.2e}" for f in PROGNOSTIC), flush=True)
    res["tolerance"][f"{case}_{dims[0]}x{dims[1]}x{dims[2]}"] = rows
    g64.close(); g32.close()

if "--skip-speed" not in sys.argv:
    wl = "basin2048"
    cs, im, jm, kb, _ = bench.WORKLOADS[wl]
    st0 = bench.build_state(wl, pdist.tile_for_rank(0, 1, im, jm))
    g0 = bench.gpu_initialise(st0, 0, None); g0.close()
    ctx = [("fp64", PomGpu(st0, device=0)), ("fp32-storage", PomGpu(st0, device=0, libpath=L.LIBPATH_F32)),
           ("fp64 ", PomGpu(st0, device=0)), ("fp32-storage ", PomGpu(st0, device=0, libpath=L.LIBPATH_F32))]
    acc = {t: {} for t, _ in ctx}
    for _, g in ctx:
        g.run(2); g.sync()
    ext = ("k_ext_", "k_advave_", "k_modeint_tail", "k_int_tail", "k_check_velocity", "k_copy2", "k_bcond1")
    for r in range(4):
        for t, g in ctx:
            g.prof_begin(); g.run(1); prof = g.prof_end()
            acc[t].setdefault("internal", []).append(sum(v[1] for k, v in prof.items() if not k.startswith(ext)))
            acc[t].setdefault("external", []).append(sum(v[1] for k, v in prof.items() if k.startswith(ext)))
            for k, v in prof.items():
                acc[t].setdefault(k[2:], []).append(v[1])
    cells = im * jm * kb
    names = ["internal", "external"] + sorted((k for k in acc["fp64"] if k not in ("internal", "external")), key=lambda k: -min(acc["fp64"][k]))[:16]
    print(f"{'min ms (2 contexts each)':26s}" + "".join(f"{t:>16s}" for t, _ in ctx))
    for n in names:
        print(f"{n:26s}" + "".join(f"{min(acc[t].get(n, [0])):16.3f}" for t, _ in ctx))
    i64 = min(min(acc["fp64"]["internal"]), min(acc["fp64 "]["internal"]))
    i32 = min(min(acc["fp32-storage"]["internal"]), min(acc["fp32-storage "]["internal"]))
    res["speed"] = {"workload": "closed basin 2048x1536x50, one MI355X, one profiled step per round, 4 rounds, 2 contexts per build",
                    "internal_ms": {"fp64": i64, "fp32-storage": i32, "ratio": i64 / i32},
                    "algorithmic_GBps": {"fp64 (133 passes x 8 B)": 133 * 8 * cells / i64 / 1e6, "fp32-storage (133 passes x 4 B)": 133 * 4 * cells / i32 / 1e6},
                    "kernel_ms_min": {n: {t.strip(): min(min(acc[t].get(n, [0])), min(acc[t2].get(n, [0]))) for t, t2 in (("fp64", "fp64 "), ("fp32-storage", "fp32-storage "))} for n in names}}
    print(json.dumps(res["speed"]["internal_ms"]), json.dumps(res["speed"]["algorithmic_GBps"]))
    for _, g in ctx:
        g.close()
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print("wrote", out_path)
