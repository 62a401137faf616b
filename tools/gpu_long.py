"""Developer check: prognostic-field drift of the HIP path against the CPU oracle over many steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from extpom_amd.cases import make_case
from extpom_amd.layout import PROGNOSTIC
from extpom_amd.model import PomGpu
from oracle.pyoracle import OracleTile, oracle_finish_initial

case = sys.argv[1] if len(sys.argv) > 1 else "seamount"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
a = make_case(case, 65, 49, 21, dte=6.0, isplit=30)
oracle_finish_initial(a)
b = a.copy()
ot = OracleTile(a)
g = PomGpu(b)
for n in [1, 10, 30, 100, 300, 1000]:
    if n > nsteps:
        break
    ot.run(n - a.iint)
    g.run(n - b.iint)
    g.download()
    r = {f: float(np.abs(a.field(f) - b.field(f)).max() / np.abs(a.field(f)).max()) for f in PROGNOSTIC + ["q2", "km", "l", "rho", "w"]}
    print(n, " ".join(f"{k}={v:.1e}" for k, v in r.items()), flush=True)
