"""The CPU oracle's steps of a LARGE grid, computed beside the GPU tests instead of inside one of them.

A step of oracle/pom_oracle.c at 2048x1536x50 takes about a minute of one core; three of them were 145 of the 180 seconds of
test_config4_2048x1536x50_full_size while the GPU sat idle.  tests/conftest.py starts this script as a child process when the
session has collected a test marked `bg_oracle(case, im, jm, kb, steps)`, the other GPU tests run meanwhile, and the marked
test joins it: after every oracle step the script leaves one digest (xxh3-128 of the 64-bit patterns) per COMMON array that is not
scratch in <outdir>/step<n>.json -- n = 0 is the initial state, so that the test can show that both sides started from identical
inputs.  Equal digests = equal bits (the comparison the test made on the arrays themselves before; what it loses is WHICH cell
differs -- the test names the arrays, `python tests/oracle_bg.py ... --keep` leaves them for a closer look).

Test infrastructure: it runs the oracle (the checker) on the host only and never opens the GPU.

    python tests/oracle_bg.py CASE IM JM KB STEPS OUTDIR
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import xxhash

from extpom_amd.layout import BLK2D, BLK3D

SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}
NML = dict(dte=6.0, isplit=30)


def digests(st):
    """one digest per compared COMMON array: the bytes of its doubles, i.e. the sign of a zero counts"""
    out = {}
    for n in BLK2D + BLK3D:
        if n in SCRATCH:
            continue
        a = np.ascontiguousarray(st.field(n), dtype="<f8")
        out[n] = xxhash.xxh3_128(memoryview(a).cast("B")).hexdigest()
    return out


def main():
    case, im, jm, kb, steps, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    from extpom_amd.cases import make_case
    from oracle.pyoracle import OracleTile, oracle_finish_initial
    os.makedirs(outdir, exist_ok=True)
    t0 = time.time()

    def leave(n, st):
        tmp = os.path.join(outdir, f".step{n}.json")
        with open(tmp, "w") as f:
            json.dump({"iint": int(st.iint), "seconds": round(time.time() - t0, 1), "digests": digests(st)}, f)
        os.replace(tmp, os.path.join(outdir, f"step{n}.json"))        # a reader never sees half a file

    a = make_case(case, im, jm, kb, **NML)
    oracle_finish_initial(a)
    leave(0, a)
    oc = OracleTile(a)
    for n in range(1, steps + 1):
        oc.run(1)
        leave(n, a)


if __name__ == "__main__":
    main()
