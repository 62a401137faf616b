#!/usr/bin/env python3
"""Developer run (one MI355X, ~5 minutes): north_star's horizon -- 1000 internal steps -- on north_star's grid, 2048x1536x50.

The CPU oracle needs 67 s per step at this size (bench.py's cpu_baseline), so 1000 steps cannot be compared with it; the tests pin steps
1-3 of this grid to the oracle and 1000 steps of smaller grids to the reference's own digests.  What CAN be shown at full size over the
whole horizon: two contexts that start from the same state and take DIFFERENT kernels through every step -- the large-grid fast paths
(two external substeps per pass marching down the rows, k_profq in 8 paced rows with its vectors in LDS, strip order of the
row-sharing kernels) against the shapes small grids use (one substep per launch, one row per wavefront, 2-row k_profq without pacing,
whole-row order), each of which is pinned to the oracle bit for bit where the oracle can follow -- hold the same bits after 100, 500
and 1000 steps, in every COMMON array that is not scratch; nothing non-finite, land masked, error_status 0, and the closed basin's
volume (area integral of et) where it started.

    python tools/fullsize_1000.py [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from extpom_amd import dist as pdist
from extpom_amd.layout import BLK2D, BLK3D, PROGNOSTIC

SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}
T0 = time.time()


def say(msg):
    print(f"{time.time() - T0:7.1f} s  {msg}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fullsize_1000_progress.log"), "a") as f:   # the GPU box's watchdog looks for signs of life
        f.write(f"{time.time() - T0:7.1f} s  {msg}\n")


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    case, im, jm, kb, desc = bench.WORKLOADS["basin2048"]
    a = bench.build_state("basin2048", pdist.tile_for_rank(0, 1, im, jm))
    ga = bench.gpu_initialise(a, 0, None)
    b = a.copy()
    from extpom_amd.model import PomGpu
    gb = PomGpu(b, device=0)
    for k, v in {"POMGPU_EXT_NOPAIR": "1", "POMGPU_EXT_NOMARCH": "1", "POMGPU_PROFQ_ROWS2": "1", "POMGPU_PROFQ_NOPACE": "1", "POMGPU_COL_STRIP": "0"}.items():
        gb.switch(k, v)
    gb.upload(b)
    area = (a.field("art") * a.fsm)
    vol0 = float((a.field("et") * area).sum())
    say(f"{desc}: two contexts, the same initial state; fast against general kernel shapes")
    done = 0
    ok = True
    for n in sorted({min(100, total), min(500, total), total}):
        t = time.time(); ga.run(n - done); ga.sync(); ta = time.time() - t
        t = time.time(); gb.run(n - done); gb.sync(); tb = time.time() - t
        ga.download(); gb.download()
        bad = [f for f in BLK2D + BLK3D if f not in SCRATCH and not np.array_equal(a.field(f).view(np.int64), b.field(f).view(np.int64))]
        fin = all(np.isfinite(a.field(f)).all() for f in PROGNOSTIC + ["q2", "km", "rho", "w"])
        land = all(not np.any((a.field(f) if a.field(f).ndim == 2 else a.field(f)[:kb - 1]) * (1.0 - a.fsm)) for f in ("t", "s", "el", "et"))
        vol = float((a.field("et") * area).sum())
        say(f"step {n}: fast {ta / (n - done) * 1e3:.2f} ms per step, general {tb / (n - done) * 1e3:.2f}; arrays that differ: {bad or 'none'}; finite {fin}; "
            f"land masked {land}; error_status {int(a.error_status)} / {int(b.error_status)}; max|u| {float(np.abs(a.field('u')).max()):.3e} "
            f"max|el| {float(np.abs(a.field('el')).max()):.3e}; sum(et*art) {vol:.6e} (start {vol0:.6e}, sum(art) {float(area.sum()):.3e})")
        ok = ok and not bad and fin and land and int(a.error_status) == 0 and int(b.error_status) == 0
        done = n
    ga.close(); gb.close()
    say("FULLSIZE-1000-OK" if ok else "FULLSIZE-1000-FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
