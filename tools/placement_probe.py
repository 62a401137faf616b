#!/usr/bin/env python3
"""Developer experiment: can the placement of a context's memory be steered?

The same library in different contexts / processes differs by up to 12 % per kernel (k_profq 7.4-7.8 ms between the
three processes of one profile job) with identical bytes and TLB misses.  Physical addresses are not visible to a user process; what a
process CAN do is change what is allocated before the context.  This run creates the bench's context several times in ONE process, each
time behind a dummy allocation of another size that stays alive meanwhile, and times the large kernels: do the times follow the dummy's
size (then a context could try a few placements at creation and keep the best), repeat for the same size, or ignore it?

    python tools/placement_probe.py [sizes in MiB ...]     the context created again and again behind dummy allocations of these sizes
    python tools/placement_probe.py hold 4                 four contexts alive at once, timed in turn
    python tools/placement_probe.py pads 0 16 65536 ...    one region, POMGPU_PAD3 = these many 4-KiB pages between the arrays of blk3d
    python tools/placement_probe.py tune                   the library's own pomgpu_tune_placement with 16 layouts, twice
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from extpom_amd import dist as pdist

KERNELS = ("k_profq", "k_ts_update", "k_advt2x2_col", "k_advq2_col", "k_advuv_col", "k_advct_col", "k_uv_filter_reg", "k_ext_pair")


def hold(n):
    """n contexts alive at once (each on other physical memory), timed in turn, twice"""
    from extpom_amd.model import PomGpu
    case, im, jm, kb, desc = bench.WORKLOADS["basin2048"]
    st = bench.build_state("basin2048", pdist.tile_for_rank(0, 1, im, jm))
    gs = [bench.gpu_initialise(st, 0, None)]
    init = st.copy()
    for _ in range(n - 1):
        gs.append(PomGpu(init.copy(), device=0))
    print(f"{'context':>10s} {'ms/step':>8s} " + " ".join(f"{k[2:12]:>10s}" for k in KERNELS), flush=True)
    for rep in range(2):
        for m, g in enumerate(gs):
            g.run(3)
            g.sync()
            g.prof_begin()
            g.run(3)
            prof = g.prof_end()
            t0 = time.perf_counter()
            g.run(10)
            g.sync()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print(f"{m:10d} {ms:8.2f} " + " ".join(f"{prof.get(k, (0, 0.0))[1] / 3:10.3f}" for k in KERNELS), flush=True)
    for g in gs:
        g.close()


def pads(values, var="POMGPU_PAD3"):
    """ONE region (a context created again in one process gets the same memory back: experiment A), another distance between the
    arrays of blk3d each time (POMGPU_PAD3, 4-KiB pages): does the layout inside the region move a kernel?"""
    from extpom_amd.model import PomGpu
    case, im, jm, kb, desc = bench.WORKLOADS["basin2048"]
    st = bench.build_state("basin2048", pdist.tile_for_rank(0, 1, im, jm))
    g = bench.gpu_initialise(st, 0, None)
    init = st.copy()
    g.close()
    print(f"{var[7:]:>10s} {'ms/step':>8s} " + " ".join(f"{k[2:12]:>10s}" for k in KERNELS), flush=True)
    for v in values:
        if v:
            os.environ[var] = str(v)
        else:
            os.environ.pop(var, None)
        g = PomGpu(init.copy(), device=0)
        g.run(3)
        g.sync()
        g.prof_begin()
        g.run(3)
        prof = g.prof_end()
        t0 = time.perf_counter()
        g.run(10)
        g.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{v:10d} {ms:8.2f} " + " ".join(f"{prof.get(k, (0, 0.0))[1] / 3:10.3f}" for k in KERNELS), flush=True)
        g.close()
        del g
        torch.cuda.empty_cache()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "hold":
        return hold(int(sys.argv[2]))
    if len(sys.argv) > 1 and sys.argv[1] == "tune":            # the library's own tuner with many candidates: the landscape of one process
        case, im, jm, kb, desc = bench.WORKLOADS["basin2048"]
        st = bench.build_state("basin2048", pdist.tile_for_rank(0, 1, im, jm))
        g = bench.gpu_initialise(st, 0, None)
        g.run(2)
        for rep in range(2):
            r = g.tune_placement(3, 16)
            print("tune:", sorted(zip(r["front_mib"], r["ms_per_step"])), "kept", r["front_mib"][r["kept"]], flush=True)
        g.close()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "pads":
        return pads([int(a) for a in sys.argv[2:]] or [0, 1, 3, 16, 64, 257, 512, 1031, 4099, 0])
    sizes = [int(a) for a in sys.argv[1:]] or [0, 0, 512, 512, 3072, 3072, 9216, 9216, 0]
    case, im, jm, kb, desc = bench.WORKLOADS["basin2048"]
    st0 = bench.build_state("basin2048", pdist.tile_for_rank(0, 1, im, jm))
    first = True
    print(f"{'dummy MiB':>10s} {'ms/step':>8s} " + " ".join(f"{k[2:12]:>10s}" for k in KERNELS), flush=True)
    for mib in sizes:
        dummy = torch.empty(mib << 20, dtype=torch.uint8, device="cuda:0") if mib else None
        st = st0 if first else st0.copy()
        g = bench.gpu_initialise(st, 0, None) if first else __import__("extpom_amd.model", fromlist=["PomGpu"]).PomGpu(st, device=0)
        if first:
            st0 = st.copy()                                   # the initialised state: later contexts upload it as it is
            first = False
        g.run(3)
        g.sync()
        g.prof_begin()
        g.run(3)
        prof = g.prof_end()
        t0 = time.perf_counter()
        g.run(10)
        g.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{mib:10d} {ms:8.2f} " + " ".join(f"{prof.get(k, (0, 0.0))[1] / 3:10.3f}" for k in KERNELS), flush=True)
        g.close()
        del g, dummy
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
