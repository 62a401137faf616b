"""Launched by tests/test_gpu_multitile.py: 2 ranks, BOTH on GPU 0 (RCCL needs one GPU per rank, so
the packed edges are staged through the host over gloo), each running the HIP path on its tile with
the C-ABI exchange hook; rank 0 then compares against the single-tile CPU oracle.
Mode "rccl" (a node with one GPU per rank): rank r on GPU r, the library's own RCCL transport between the ranks
(extpom_amd.halo.connect_rccl, both streams and communicators) and the wide-halo external mode -- bench.py's N > 1 path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from extpom_amd import decomp
from extpom_amd.cases import finish_initial, make_case
from extpom_amd.halo import DeviceHalo, Halo
from extpom_amd.layout import BLK2D, BLK3D

# POM_TILES_STEPS / POM_TILES_GRID ("384x256x24") / POM_TILES_ISPLIT: longer and larger soak runs
IM, JM, KB = (int(v) for v in os.environ.get("POM_TILES_GRID", "97x61x16").split("x"))
STEPS = int(os.environ.get("POM_TILES_STEPS", "3"))
ISPLIT = int(os.environ.get("POM_TILES_ISPLIT", "10"))
SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}


def worker(rank, world, split, port, out, nml):
    nml = dict(nml)
    mode = nml.pop("_exchange", "hook")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = rank if mode == "rccl" else 0
    torch.cuda.set_device(dev)
    from extpom_amd.model import PomGpu
    nx, ny = {"x": (2, 1), "y": (1, 2), "xy": (2, 2), "y4": (1, 4)}[split]
    iml, jml = decomp.local_size(IM, JM, nx, ny)
    tile = decomp.make_tile(rank, IM, JM, iml, jml, n_proc=world)
    st = make_case("island", IM, JM, KB, tile=tile, dte=6.0, isplit=ISPLIT, **nml)
    # kernels and torch's pack/unpack must share ONE stream; torch's default stream has handle 0, which
    # the C ABI reads as "create your own", so make a real stream current and hand that over
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    g = PomGpu(st, device=dev, stream=ts.cuda_stream)
    if mode == "hook":
        halo = DeviceHalo(g, tile, torch.device("cuda", 0), staged=True)
        g.set_order_exchange(Halo(tile, staged=True).device_order_hook(torch.device("cuda", 0)))   # baropg_mcc (npg = 2)
    else:                                        # the library serves the exchange points itself (pomgpu_set_transport)
        from extpom_amd.halo import StagedMover, connect_rccl
        if mode == "rccl":
            assert connect_rccl(g, tile, rank, world), "the RCCL transport could not connect the ranks"
        else:
            from extpom_amd.halo import dist_allmin
            g.set_transport(tile, StagedMover(g, tile, torch.device("cuda", 0)), agree=dist_allmin())
        if mode in ("wide", "rccl"):
            tiles = [decomp.make_tile(r, IM, JM, iml, jml, n_proc=world) for r in range(world)]
            assert g.set_wide_external(True, min(t.im for t in tiles), min(t.jm for t in tiles))

    def dens(s, a, b, c):
        g.upload(s); g.call("dens", a, b, c); g.download(s)

    def baropg(s):
        g.upload(s); g.call("baropg_mcc" if int(s.npg) == 2 else "baropg"); g.download(s)

    finish_initial(st, dens, baropg)
    g.upload(st)
    g.prof_begin()
    g.run(STEPS)
    kernels = ",".join(sorted(g.prof_end()))
    # the output file of the reference, every rank its patch (rank 0 lays the file out first)
    if rank == 0:
        g.write_file("output", os.path.join(out, "out.nc"), title="tiles", time_start="t0", im_global=IM, jm_global=JM, create=True)
    dist.barrier()
    if rank != 0:
        g.write_file("output", os.path.join(out, "out.nc"), title="tiles", time_start="t0", im_global=IM, jm_global=JM, create=False)
    dist.barrier()
    g.download()
    np.savez(os.path.join(out, f"tile{rank}.npz"), i_off=tile.i_off, j_off=tile.j_off, im=tile.im, jm=tile.jm,
             n=(halo.count if mode == "hook" else g.exchange_rounds() + g.exchange_rounds_side()), kernels=np.array(kernels), **{n: st.field(n) for n in BLK2D + BLK3D if n not in SCRATCH})
    g.close()
    dist.barrier()
    dist.destroy_process_group()


def main(split, nml, exchange="hook"):
    import tempfile
    from oracle.pyoracle import OracleTile, oracle_finish_initial
    out = tempfile.mkdtemp()
    port = 29700 + (os.getpid() % 200)
    world = 4 if split in ("xy", "y4") else 2
    mp.spawn(worker, args=(world, split, port, out, dict(nml, _exchange=exchange)), nprocs=world, join=True)
    g = make_case("island", IM, JM, KB, dte=6.0, isplit=ISPLIT, **nml)
    oracle_finish_initial(g)
    OracleTile(g).run(STEPS)
    bad = []
    for r in range(world):
        z = np.load(os.path.join(out, f"tile{r}.npz"))
        io, jo, im, jm = int(z["i_off"]), int(z["j_off"]), int(z["im"]), int(z["jm"])
        assert int(z["n"]) > (15 if exchange in ("wide", "rccl") else 100), int(z["n"])
        if r == 0:
            print("message rounds on rank 0:", int(z["n"]))
            print("kernels on rank 0:", str(z["kernels"]))
        sl_j = slice(0 if jo == 0 else 1, jm if jo + jm == JM else jm - 1)
        sl_i = slice(0 if io == 0 else 1, im if io + im == IM else im - 1)
        for n in BLK2D + BLK3D:
            if n in SCRATCH:
                continue
            ref = g.field(n)[..., jo:jo + jm, io:io + im][..., sl_j, sl_i]
            got = z[n][..., :jm, :im][..., sl_j, sl_i]
            if not np.array_equal(ref, got):
                bad.append((r, n, float(np.abs(ref - got).max())))
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(out, "out.nc"), "r", mmap=False) as f:     # one file, written by all ranks
        for n in ("t", "s", "u", "v", "rho"):
            if not np.array_equal(f.variables[n][0], g.field(n)[:KB - 1]):
                bad.append(("file", n, 0.0))
        for n in ("elb", "uab", "vab"):
            if not np.array_equal(f.variables[n][0], g.field(n)):
                bad.append(("file", n, 0.0))
        if not np.array_equal(f.variables["h"][:], g.field("h")):
            bad.append(("file", "h", 0.0))
    if bad:
        print("MISMATCH", bad[:20])
        sys.exit(1)
    print("TILES-OK", split)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "x", dict(npg=2) if "npg2" in sys.argv[2:] else {},
         "rccl" if "rccl" in sys.argv[2:] else ("wide" if "wide" in sys.argv[2:] else ("transport" if "transport" in sys.argv[2:] else "hook")))
