"""Launched by tests/test_oracle_golden.py (cpu) and tests/test_gpu_multitile.py (gpu): four ranks, a 2x2 split of the
65x49x21 case with the reference's own tile size (34 x 26); every rank steps its tile -- the CPU oracle with
extpom_amd.halo.Halo as its exchange (`cpu`), or the HIP path, all four ranks on GPU 0, with the per-point hooks, the
library exchange or the wide-halo external mode (`hook` / `transport` / `wide`) -- and compares the SHA-256 of every
restart-list field over its (jm, im) cells, GHOST CELLS INCLUDED, with what the REFERENCE ITSELF left on that rank in
a four-process MPICH run (tests/golden/tiles_65x49x21_2x2.json, tests/golden/make_golden_tiles.py)."""
import hashlib
import json
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
from extpom_amd.halo import Halo

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "tiles_65x49x21_2x2.json")))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<f8").tobytes()).hexdigest()


def worker(rank, world, port, out, name, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    (IM, JM, KB), (IML, JML) = GOLD["grid"], GOLD["local"]
    cfg = GOLD["configs"][name]
    tile = decomp.make_tile(rank, IM, JM, IML, JML, n_proc=world)
    st = make_case(cfg["case"], IM, JM, KB, tile=tile, **cfg["nml"])
    if mode == "cpu":
        from oracle.pyoracle import OracleTile
        halo = Halo(tile)
        ot = OracleTile(st, exch2d=halo.numpy_hook2d(), exch3d=halo.numpy_hook3d(), order=halo.numpy_order_hook())
        finish_initial(st, lambda s, a, b, c: ot.call("dens", ot.a3(a), ot.a3(b), ot.a3(c)),
                       lambda s: ot.call("baropg_mcc" if int(s.npg) == 2 else "baropg"))
        run, sync = ot.run, (lambda: None)
    else:
        from extpom_amd.halo import DeviceHalo, StagedMover
        from extpom_amd.model import PomGpu
        torch.cuda.set_device(0)
        ts = torch.cuda.Stream()
        torch.cuda.set_stream(ts)
        g = PomGpu(st, device=0, stream=ts.cuda_stream)
        dev = torch.device("cuda", 0)
        if mode == "hook":
            keep = DeviceHalo(g, tile, dev, staged=True)      # noqa: F841
            g.set_order_exchange(Halo(tile, staged=True).device_order_hook(dev))
        else:
            from extpom_amd.halo import dist_allmin
            g.set_transport(tile, StagedMover(g, tile, dev), agree=dist_allmin())
            if mode == "wide":
                tiles = [decomp.make_tile(r, IM, JM, IML, JML, n_proc=world) for r in range(world)]
                assert g.set_wide_external(True, min(t.im for t in tiles), min(t.jm for t in tiles))

        def dens(s, a, b, c):
            g.upload(s); g.call("dens", a, b, c); g.download(s)

        def baropg(s):
            g.upload(s); g.call("baropg_mcc" if int(s.npg) == 2 else "baropg"); g.download(s)

        finish_initial(st, dens, baropg)
        g.upload(st)
        run, sync = g.run, g.download
    bad, done = [], 0
    for step in sorted(int(s) for s in cfg["steps"]):
        run(step - done)
        done = step
        sync()
        want = cfg["steps"][str(step)][rank]
        for f in GOLD["fields"]:
            if digest(st.field(f)[..., :tile.jm, :tile.im]) != want[f]:
                bad.append((step, f))
    if mode == "wide":      # on the library's second stream: four rounds per internal step and five more from the second step on (pomgpu_api.hip, "rim rounds")
        assert g.exchange_rounds_side() == 9 * done - 5, (g.exchange_rounds_side(), done)
    with open(os.path.join(out, f"rank{rank}.json"), "w") as fh:
        json.dump(bad, fh)
    dist.barrier()
    dist.destroy_process_group()


def main(name, mode):
    import tempfile
    out = tempfile.mkdtemp()
    port = 29900 + (os.getpid() % 90)
    mp.spawn(worker, args=(4, port, out, name, mode), nprocs=4, join=True)
    bad = {r: json.load(open(os.path.join(out, f"rank{r}.json"))) for r in range(4)}
    if any(bad.values()):
        print("MISMATCH", {r: b[:12] for r, b in bad.items() if b})
        sys.exit(1)
    print("TILES-GOLDEN-OK", name, mode)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "cpu")
