"""BASELINE configs[4], the tolerance half on the config's OWN grid: libpomgpu.so and libpomgpu_f32.so (the fp32-storage study variant)
stepped side by side in one process from identical initial states of a bench workload; after each checkpoint the largest difference of
every prognostic field relative to the field's largest magnitude.  Used by tools/fp32_study_gpu.py (--full-drift) and by the GPU test
that asserts the 10-step envelope at 2048x1536x50."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np

import bench
from extpom_amd import dist as pdist, lib as L
from extpom_amd.layout import PROGNOSTIC
from extpom_amd.model import PomGpu


def rel(a, b, f):
    x, y = a.field(f), b.field(f)
    return float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-300))


def full_size_drift(steps_list, fields=PROGNOSTIC, workload="basin2048", beat=lambda m: None):
    """both builds from the same initial state of the bench grid; only the compared fields come back to the host"""
    import ctypes
    from extpom_amd.layout import P2, P3
    cs, im, jm, kb, desc = bench.WORKLOADS[workload]
    a = bench.build_state(workload, pdist.tile_for_rank(0, 1, im, jm))
    g0 = bench.gpu_initialise(a, 0, None); g0.close()
    b = a.copy()
    beat("states built")
    g64, g32 = PomGpu(a, device=0), PomGpu(b, device=0, libpath=L.LIBPATH_F32)

    def fetch(g, st):
        for f in fields:
            dst = ctypes.c_void_p(st.field(f).ctypes.data)
            g._chk((g.L.pomgpu_download_3d if f in P3 else g.L.pomgpu_download_2d)(g.h, (P3 if f in P3 else P2)[f], dst), "download " + f)
    done, rows = 0, {}
    for n in steps_list:
        g64.run(n - done); g32.run(n - done); done = n
        fetch(g64, a); fetch(g32, b)
        g64.get_con(); g32.get_con()
        rows[str(n)] = {f: rel(a, b, f) for f in fields}
        rows[str(n)]["largest_magnitude_fp64"] = {f: float(np.abs(a.field(f)).max()) for f in fields}
        rows[str(n)]["error_status"] = [int(a.error_status), int(b.error_status)]
        beat(f"step {n} compared")
        print(f"{workload} {im}x{jm}x{kb} step {n:5d}: " + "  ".join(f"{f}={rows[str(n)][f]:.2e}" for f in fields), flush=True)
    out = {"workload": desc, "builds": {"fp64": g64.L.pomgpu_version().decode(), "fp32-storage": g32.L.pomgpu_version().decode()},
           "what": "largest |fp32-storage - fp64| of a field after n internal steps from identical initial states, relative to the field's largest magnitude (fp64 run)",
           "steps": rows}
    g64.close(); g32.close()
    return out


