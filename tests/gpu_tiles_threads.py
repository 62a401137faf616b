"""Launched by tests/test_gpu_multitile.py: decomposition invariance AT FULL SIZE, GPU against GPU, in one process.

The reference guarantees that a run does not depend on how the domain is cut into tiles (its only
regression check).  Here the bench grid (default 2048x1536x50, the grid north_star's targets are stated on) runs once as ONE
tile and once as A x B tiles -- 1 x N whole rows (bench.py's default split) or the reference's own kind of 2-D split, e.g. the 2 x 4
of BASELINE configs[2] / [3] with its trimmed north row of tiles and all eight neighbours (parallel_mpi.f:54-65,82-119,96-103) --
all on GPU 0: every tile a context of its own on a stream of its own, driven by a host thread, the library's exchange
(pomgpu_set_transport) with a mover that copies the staging buffers device to device between the contexts (corner messages
included), the wide-halo external mode with its rounds on the second stream.  After STEPS internal steps every cell a tile OWNS must hold the bits of the single-tile run, in every COMMON array
that is not pure scratch.  The single-tile path itself is pinned to the oracle at this size for steps 1-3
(test_config4_2048x1536x50_full_size); this carries that pin over STEPS steps and over the multi-tile code path.

    python tests/gpu_tiles_threads.py [IMxJMxKB] [N | AxB[,CxD...]] [STEPS] [f32]
"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from extpom_amd import decomp
from extpom_amd import lib as _lib
from extpom_amd.cases import finish_initial, make_case
from extpom_amd.halo import _DevPtr
from extpom_amd.layout import BLK2D, BLK3D
from extpom_amd.lib import OPP
from extpom_amd.model import PomGpu

SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}
# fp32-storage variant on tiles: the envelope of its difference to the single tile after a few steps (fp32 rounding is 6e-8; the
# temperature and salinity fields carry it directly, the elevation through the pressure gradient)
F32_BOUND = {"t": 1e-5, "s": 1e-5, "tb": 1e-5, "sb": 1e-5, "rho": 1e-4, "u": 5e-2, "v": 5e-2, "ub": 5e-2, "vb": 5e-2, "el": 5e-2, "elb": 5e-2, "et": 5e-2, "ua": 5e-2, "va": 5e-2}
T0 = time.time()


def beat(msg):                                       # the GPU box's watchdog looks for signs of life under gpurun_out/
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/tiles_threads_progress.log", "a") as f:
        f.write(f"{time.time() - T0:7.1f} s  {msg}\n")


class Board:
    def __init__(self, world):
        self.box = {}
        self.barrier = threading.Barrier(world)

    def allmin(self, me, value):
        self.box[("min", me)] = int(value)
        self.barrier.wait()
        m = min(v for k, v in self.box.items() if k[0] == "min")
        self.barrier.wait()
        return m


def gpu_finish(st, g):
    def dens(s, a, b, c):
        g.upload(s); g.call("dens", a, b, c); g.download(s)

    def baropg(s):
        g.upload(s); g.call("baropg_mcc" if int(s.npg) == 2 else "baropg"); g.download(s)

    finish_initial(st, dens, baropg)
    g.upload(st)


def main():
    grid = sys.argv[1] if len(sys.argv) > 1 else "2048x1536x50"
    splits = [tuple(int(v) for v in sp.split("x")) if "x" in sp else (1, int(sp)) for sp in (sys.argv[2] if len(sys.argv) > 2 else "4").split(",")]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    libpath = _lib.LIBPATH_F32 if "f32" in sys.argv[4:] else None
    # "stored_only": every context is created under POMGPU_SUM2D_OFF (the vertical integrals of advance.f:152-168 formed from the STORED arrays by
    # k_vint instead of from the registers of advct / baropg) and POMGPU_QFILTER_SPLIT (the Asselin filter of q2, q2l as a kernel of its own that
    # reads the stored q2f, instead of inside k_profq's walk up, which holds it in fp64).  In fp64 neither changes a bit.  In the fp32-storage
    # variant they remove the places where tiles and the single tile round differently -- a tile's first and last owned rows are filtered by the
    # edge-line kernel (stored values) where the single tile's fused kernel has the unrounded ones, and two lines of adx2d, ady2d are re-summed
    # from stored values after the neighbour's edge lines have arrived -- and with them the variant is decomposition-invariant bit for bit
    exact = libpath is None or "stored_only" in sys.argv[4:]
    if "stored_only" in sys.argv[4:]:
        os.environ["POMGPU_SUM2D_OFF"] = "1"
        os.environ["POMGPU_QFILTER_SPLIT"] = "1"
    im, jm, kb = (int(v) for v in grid.split("x"))
    nml = dict(dte=6.0, isplit=30, mode=3, nadv=2, nitera=1, npg=1)
    dev = torch.device("cuda", 0)
    # ---- one tile ----------------------------------------------------------------------------------------------------
    a = make_case("basin", im, jm, kb, **nml)
    ga = PomGpu(a, device=0, libpath=libpath)
    gpu_finish(a, ga)
    ga.run(steps)
    ga.download()
    ga.close()
    assert a.error_status == 0
    beat(f"single tile: {steps} steps done")
    for nx, ny in splits:                             # every split against the same single-tile run
        run_split(a, grid, im, jm, kb, nml, nx, ny, steps, libpath, dev, exact)


def run_split(a, grid, im, jm, kb, nml, nx, ny, steps, libpath, dev, exact):
    world = nx * ny
    # ---- nx x ny tiles, one host thread each -------------------------------------------------------------------------
    iml, jml = decomp.local_size(im, jm, nx, ny)
    assert decomp.tile_grid(im, jm, iml, jml) == (nx, ny), (decomp.tile_grid(im, jm, iml, jml), nx, ny)
    tiles = [decomp.make_tile(r, im, jm, iml, jml, n_proc=world) for r in range(world)]
    print("tiles (im x jm, neighbours W E S N SW SE NW NE):", [(t.im, t.jm, PomGpu.neighbours8(t)) for t in tiles])
    board, errs, bad, info, worst = Board(world), [], [], {}, {}

    def rank(r):
        try:
            torch.cuda.set_device(0)
            tile = tiles[r]
            st = make_case("basin", im, jm, kb, tile=tile, **nml)
            ts = torch.cuda.Stream()
            torch.cuda.set_stream(ts)                 # thread-local: this thread's torch work goes to its context's stream
            g = PomGpu(st, device=0, stream=ts.cuda_stream, libpath=libpath)
            nb = PomGpu.neighbours8(tile)
            w = lambda p, n: torch.as_tensor(_DevPtr(p, (n,)), device=dev)

            def mover(send, scount, recv, rcount):
                # ASYNCHRONOUS, like the RCCL transport: nothing here waits for the device.  The copies are enqueued on the stream of the
                # round (pomgpu_current_stream: the kernels' stream or the library's second one) behind an event the neighbour recorded
                # after its pack; the host threads only meet at two barriers to hand the events over.  The rounds on the second stream
                # therefore really run beside the kernels of the first, and a main-stream kernel that does not wait for the round it
                # depends on reads stale ghost cells here as it would between GPUs.
                cs = torch.cuda.ExternalStream(g.current_stream())
                packed = torch.cuda.Event()
                packed.record(cs)
                for d in range(8):
                    if nb[d] >= 0 and scount[d]:
                        board.box[(r, nb[d], d)] = (send[d], scount[d], packed)
                board.barrier.wait()
                with torch.cuda.stream(cs):
                    for d in range(8):
                        if nb[d] >= 0 and rcount[d]:
                            p, n, ev = board.box[(nb[d], r, OPP[d])]
                            assert n == rcount[d], (r, d, n, rcount[d])
                            cs.wait_event(ev)                  # the neighbour's pack kernel has filled its staging buffer
                            w(recv[d], n).copy_(w(p, n), non_blocking=True)
                taken = torch.cuda.Event()
                taken.record(cs)
                board.box[("taken", r)] = taken
                board.barrier.wait()
                for d in range(8):                            # nobody repacks a buffer a neighbour is still reading: this stream's next pack waits
                    if nb[d] >= 0 and scount[d]:
                        cs.wait_event(board.box[("taken", nb[d])])
                board.barrier.wait()                          # ... and nobody overwrites a mailbox entry a neighbour has yet to read

            g.set_transport(tile, mover, agree=lambda mine: board.allmin(r, mine), stream_ordered=True)
            assert g.set_wide_external(True, min(t.im for t in tiles), min(t.jm for t in tiles))
            gpu_finish(st, g)
            board.barrier.wait()
            g.run(steps)
            g.download()
            info[r] = (g.exchange_rounds(), g.exchange_rounds_side())
            g.close()
            assert st.error_status == 0
            io, jo, ti, tj = tile.i_off, tile.j_off, tile.im, tile.jm
            sl_j = slice(0 if jo == 0 else 1, tj if jo + tj == jm else tj - 1)
            sl_i = slice(0 if io == 0 else 1, ti if io + ti == im else ti - 1)
            for n in BLK2D + BLK3D:
                if n in SCRATCH:
                    continue
                ref = np.ascontiguousarray(a.field(n)[..., jo:jo + tj, io:io + ti][..., sl_j, sl_i])
                got = np.ascontiguousarray(st.field(n)[..., :tj, :ti][..., sl_j, sl_i])
                if exact:
                    if not np.array_equal(ref.view(np.int64), got.view(np.int64)):     # the bits, the sign of a zero included
                        bad.append((r, n, float(np.abs(ref - got).max())))
                        if os.environ.get("POM_TILES_VERBOSE"):                         # developer: where (local indices of the compared block) and how many
                            d = np.argwhere(ref.view(np.int64) != got.view(np.int64))
                            print(f"  tile {r} {n}: {len(d)} cells differ, index ranges {d.min(axis=0).tolist()} .. {d.max(axis=0).tolist()}, first {d[0].tolist()}", flush=True)
                else:
                    # the fp32-storage study variant is NOT decomposition-invariant bit for bit: a fused kernel integrates the values it has
                    # in registers (fp64) where the tile path's edge-line kernels re-read them from memory (rounded to fp32) -- storage
                    # rounding enters at other places, and the flow amplifies it like any other fp32-level difference
                    rel = float(np.abs(ref - got).max() / max(float(np.abs(a.field(n)).max()), 1e-300))
                    worst[n] = max(worst.get(n, 0.0), rel)
                    if n in F32_BOUND and rel > F32_BOUND[n]:
                        bad.append((r, n, rel))
            beat(f"{nx}x{ny}: tile {r} compared")
        except Exception as e:                        # noqa: BLE001 -- a dead rank must not leave the others at the barrier
            import traceback
            errs.append(traceback.format_exc())
            board.barrier.abort()

    threads = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        print("ERROR", errs[0])
        sys.exit(2)
    print(f"message rounds per step and tile: {[round(v[0] / steps, 2) for v in info.values()]} between kernels, "
          f"{[round(v[1] / steps, 2) for v in info.values()]} on the second stream")
    if worst:
        print("fp32-storage variant, tiles against one tile, largest difference relative to the field's largest magnitude:",
              {k: float(f"{v:.2e}") for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:12]})
    if bad:
        print("MISMATCH", bad[:20])
        sys.exit(1)
    print(f"TILES-THREADS-OK {grid} {nx}x{ny} {steps} steps")


if __name__ == "__main__":
    main()
