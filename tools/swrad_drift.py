#!/usr/bin/env python3
"""developer measurement: proft with nbct = 2 and swrad != 0 -- the reference evaluates its two exp() in REAL(16) (solver.f:1608-1611),
the kernel in fp64 -- how far are the prognostic fields from the oracle after 5 ... 1000 internal steps (65x49x21 seamount)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from extpom_amd.cases import make_case
from extpom_amd.layout import PROGNOSTIC
from extpom_amd.model import PomGpu
from oracle.pyoracle import OracleTile, oracle_finish_initial

for nbct in (2, 4):
    a = make_case("seamount", 65, 49, 21, dte=6.0, isplit=30, nbct=nbct)
    a.swrad[...] = -5.0e-5 * a.fsm
    oracle_finish_initial(a)
    b = a.copy()
    oa, g = OracleTile(a), PomGpu(b, device=0)
    done = 0
    for n in (5, 20, 100, 300, 1000):
        oa.run(n - done); g.run(n - done); done = n
        g.download()
        r = {f: float(np.abs(a.field(f) - b.field(f)).max() / max(np.abs(a.field(f)).max(), 1e-300)) for f in PROGNOSTIC}
        print(f"nbct={nbct} step {n}: " + " ".join(f"{k}={v:.2e}" for k, v in r.items()), flush=True)
    g.close()
