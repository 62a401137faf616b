"""Launched by tests/test_host_logic.py: spawns 2 gloo ranks; each runs the CPU oracle on its tile
with extpom_amd.halo.Halo as the exchange, then rank 0 compares with the single-tile run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from extpom_amd import decomp
from extpom_amd.cases import make_case
from extpom_amd.halo import Halo
from extpom_amd.layout import BLK2D, BLK3D
from oracle.pyoracle import OracleTile, oracle_finish_initial

IM, JM, KB, STEPS = 41, 35, 11, 4
SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}


def worker(rank, world, split, port, out, nml):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nx, ny = {"x": (2, 1), "y": (1, 2), "xy": (2, 2)}[split]
    iml, jml = decomp.local_size(IM, JM, nx, ny)
    tile = decomp.make_tile(rank, IM, JM, iml, jml, n_proc=world)
    st = make_case("island", IM, JM, KB, tile=tile, dte=6.0, isplit=10, **nml)
    halo = Halo(tile)
    ot = OracleTile(st, exch2d=halo.numpy_hook2d(), exch3d=halo.numpy_hook3d(), order=halo.numpy_order_hook())
    from extpom_amd.cases import finish_initial
    finish_initial(st, lambda s, a, b, c: ot.call("dens", ot.a3(a), ot.a3(b), ot.a3(c)),
                   lambda s: ot.call("baropg_mcc" if int(s.npg) == 2 else "baropg"))
    ot.run(STEPS)
    np.savez(os.path.join(out, f"tile{rank}.npz"), i_off=tile.i_off, j_off=tile.j_off, im=tile.im, jm=tile.jm,
             n=halo.count, **{n: st.field(n) for n in BLK2D + BLK3D if n not in SCRATCH})
    dist.barrier()
    dist.destroy_process_group()


def main(split, nml):
    import tempfile
    out = tempfile.mkdtemp()
    port = 29600 + (os.getpid() % 200)
    world = 4 if split == "xy" else 2
    mp.spawn(worker, args=(world, split, port, out, nml), nprocs=world, join=True)
    g = make_case("island", IM, JM, KB, dte=6.0, isplit=10, **nml)
    oracle_finish_initial(g)
    OracleTile(g).run(STEPS)
    bad = []
    for r in range(world):
        z = np.load(os.path.join(out, f"tile{r}.npz"))
        io, jo, im, jm = int(z["i_off"]), int(z["j_off"]), int(z["im"]), int(z["jm"])
        assert int(z["n"]) > 100
        for n in BLK2D + BLK3D:
            if n in SCRATCH:
                continue
            ref = g.field(n)[..., jo:jo + jm, io:io + im]
            got = z[n][..., :jm, :im]
            # the reference leaves a few work arrays' ghost cells stale between exchanges; compare the
            # cells a tile OWNS (its interior plus physical edges)
            sl_j = slice(0 if jo == 0 else 1, jm if jo + jm == JM else jm - 1)
            sl_i = slice(0 if io == 0 else 1, im if io + im == IM else im - 1)
            if not np.array_equal(ref[..., sl_j, sl_i], got[..., sl_j, sl_i]):
                bad.append((r, n, float(np.abs(ref[..., sl_j, sl_i] - got[..., sl_j, sl_i]).max())))
    if bad:
        print("MISMATCH", bad[:20])
        sys.exit(1)
    print("HALO-OK", split)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "x", dict(npg=2) if "npg2" in sys.argv[2:] else {})
