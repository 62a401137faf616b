#!/usr/bin/env python3
"""BASELINE configs[4] asks what an fp32 internal mode would cost in accuracy.  This tool answers the storage half of
that question on the CPU oracle (no GPU): the same run twice, once in fp64 throughout and once with the 3-D prognostic
fields rounded to fp32 after every internal step (what keeping them in fp32 in HBM would do at the very least --
fp32 ARITHMETIC would add to it), and prints the relative difference of the prognostic fields as the run goes on.

    python tools/fp32_storage_study.py [case] [im jm kb] [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from extpom_amd.cases import make_case
from oracle.pyoracle import OracleTile, oracle_finish_initial

PROG3 = ("u", "ub", "v", "vb", "t", "tb", "s", "sb", "q2", "q2b", "q2l", "q2lb", "km", "kh", "kq", "rho")
SHOW = ("u", "v", "t", "s", "el")


def main():
    a = sys.argv[1:]
    case = a[0] if a else "seamount"
    im, jm, kb = (int(a[1]), int(a[2]), int(a[3])) if len(a) > 3 else (65, 49, 21)
    steps = int(a[4]) if len(a) > 4 else 1000
    ref = make_case(case, im, jm, kb, dte=6.0, isplit=30)
    oracle_finish_initial(ref)
    low = ref.copy()
    o64, o32 = OracleTile(ref), OracleTile(low)
    marks = sorted({1, 10, 100, 300, 1000, steps} & set(range(1, steps + 1)))
    print("step  " + "  ".join(f"{n:>10s}" for n in SHOW) + "   (max |fp32-stored - fp64| / max |fp64|)")
    for n in range(1, steps + 1):
        o64.run(1)
        o32.run(1)
        for f in PROG3:
            x = low.field(f)
            x[...] = x.astype(np.float32).astype(np.float64)
        if n in marks:
            d = [float(np.abs(low.field(f) - ref.field(f)).max() / max(np.abs(ref.field(f)).max(), 1e-300)) for f in SHOW]
            print(f"{n:5d} " + "  ".join(f"{v:10.3e}" for v in d), flush=True)


if __name__ == "__main__":
    main()
