"""Developer check on a GPU box: HIP path vs CPU oracle on several small cases + a first timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from extpom_amd.cases import make_case
from extpom_amd.layout import BLK2D, BLK3D
from extpom_amd.model import PomGpu
from oracle.pyoracle import OracleTile, oracle_finish_initial

SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}


def reldiff(a, b):
    worst = (0.0, None)
    for n in BLK2D + BLK3D:
        if n in SCRATCH:
            continue
        x, y = a.field(n), b.field(n)
        s = np.abs(x).max()
        d = np.abs(x - y).max() / s if s > 0 else np.abs(y).max()
        if not np.isfinite(d):
            d = np.inf
        if d > worst[0]:
            worst = (float(d), n)
    return worst


def run_case(case, steps, **kw):
    a = make_case(case, 65, 49, 21, dte=6.0, isplit=30, **kw)
    oracle_finish_initial(a)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b)
    out = []
    for n in steps:
        done = a.iint
        ot.run(n - done)
        g.run(n - done)
        g.download()
        out.append((n,) + reldiff(a, b))
    print(case, kw, out, "vamax", g.check_velocity(), ot.vamax, flush=True)
    g.close()


def timing(case, im, jm, kb, nsteps):
    a = make_case(case, im, jm, kb, dte=6.0, isplit=30)
    t0 = time.time()
    oracle_finish_initial(a)
    g = PomGpu(a)
    g.run(2)
    g.sync()
    t0 = time.time()
    g.run(nsteps)
    g.sync()
    dt = (time.time() - t0) / nsteps
    g.prof_begin()
    g.run(2)
    prof = g.prof_end()
    g.download()
    print(f"{case} {im}x{jm}x{kb}: {dt*1e3:.3f} ms/step  {im*jm*kb/dt/1e9:.3f} Gcell/s  err={a.error_status} max|u|={np.abs(a.u).max():.3f}", flush=True)
    tot = sum(ms for _, ms in prof.values())
    for name, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {name:22s} n={n:5d} {ms/2:9.3f} ms/step  {100*ms/tot:5.1f}%")
    print(f"   sum of kernel time {tot/2:.3f} ms/step")
    g.close()


if __name__ == "__main__":
    run_case("seamount", [1, 2, 3, 10, 60])
    run_case("island", [3, 20])
    run_case("basin", [3, 20])
    run_case("seamount", [5], nadv=1)
    run_case("island", [5], nitera=2)
    run_case("seamount", [5], mode=4)
    run_case("seamount", [5], mode=2)
    run_case("seamount", [5], nbct=3, nbcs=3)
    timing("seamount", 65, 49, 21, 20)
    timing("seamount", 256, 256, 30, 20)
    timing("basin", 1024, 1024, 40, 5)
