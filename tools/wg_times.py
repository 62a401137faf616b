#!/usr/bin/env python3
"""Developer measurement: when and where does each workgroup of k_advct_col run?

Needs a library built with -DPOMGPU_WGTIME (tools/build_variant.sh wgtime -DPOMGPU_WGTIME): every workgroup of k_advct_col then
leaves its start and end time (wall_clock64, 100 MHz) and its hardware ids in `wr`.  The question behind it: a
launch on a 194-row tile is 3.3 rounds of CU-filling workgroups and costs 4.3 round times -- where does the extra round sit?

    POMGPU_LIBPATH=build_variants/libpomgpu_wgtime.so python tools/wg_times.py [rows ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from extpom_amd.cases import make_case
from extpom_amd.model import PomGpu


def one(jm):
    im, kb = 2048, 50
    st = make_case("basin", im, jm, kb, **bench.NML)
    g = bench.gpu_initialise(st, 0, None)
    g.run(2)
    for _ in range(3):
        g.call("advct")
    g.sync()
    g.download()
    nby = (jm + 7) // 8
    nwg = 8 * ((nby + 7) // 8) * 34                                # the larger of the two workgroup orders' grids
    rec = st.field("wr").reshape(-1)[:4 * (nwg + 64)].reshape(-1, 4)
    ok = (rec[:, 1] > rec[:, 0]) & (rec[:, 0] > 0)
    rec = rec[ok]
    rec = rec[rec[:, 1] > rec[:, 1].max() - 2.0e5]                 # the LAST launch only (2 ms back; earlier launches wrote other wr slots)
    t0, t1 = rec[:, 0] / 100.0, rec[:, 1] / 100.0                  # us
    hw, xcc = rec[:, 2].astype(np.int64), rec[:, 3].astype(np.int64) & 0xF
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)   # xcc, se, sh, cu
    base = t0.min()
    t0 -= base; t1 -= base
    dur = t1 - t0
    order = np.argsort(t0)
    n = len(rec)
    ncu = len(set(cu.tolist()))
    print(f"--- 2048x{jm}x{kb}: {n} workgroups on {ncu} distinct CUs; launch spans {t1.max():.1f} us; workgroup duration min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f} us")
    # by start order: the first `ncu` workgroups are the first round
    for a, b, name in ((0, ncu, "first round"), (ncu, 2 * ncu, "second round"), (2 * ncu, 3 * ncu, "third round"), (3 * ncu, n, "rest")):
        idx = order[a:b]
        if len(idx):
            print(f"    {name:13s} {len(idx):4d} workgroups: start {t0[idx].min():7.1f} .. {t0[idx].max():7.1f} us, duration mean {dur[idx].mean():6.1f} (min {dur[idx].min():.1f}, max {dur[idx].max():.1f}), end {t1[idx].min():7.1f} .. {t1[idx].max():7.1f}")
    # per CU: busy time and gaps
    busy, gaps, per = [], [], []
    for c in set(cu.tolist()):
        m = cu == c
        s, e = t0[m], t1[m]
        o = np.argsort(s)
        s, e = s[o], e[o]
        per.append(len(s))
        busy.append(float((e - s).sum()))
        gaps += [float(s[k + 1] - e[k]) for k in range(len(s) - 1)]
    span = t1.max()
    print(f"    per CU: {min(per)}-{max(per)} workgroups, busy {np.mean(busy) / span * 100:.0f} % of the launch on average (min {min(busy) / span * 100:.0f} %); gap between two workgroups on a CU: "
          f"median {np.median(gaps) if gaps else 0:.1f} us, max {max(gaps) if gaps else 0:.1f}")
    # how many workgroups are running at time t
    ts = np.linspace(0, span, 21)
    act = [(int(((t0 <= t) & (t1 > t)).sum())) for t in ts]
    print("    running workgroups at 0, 5, ... 100 % of the launch:", act)
    g.close()


if __name__ == "__main__":
    for jm in ([int(a) for a in sys.argv[1:]] or [194, 386, 1536]):
        one(jm)
