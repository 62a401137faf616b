"""Launched by tests/test_gpu_multitile.py: ONE rank, its tile its own western and eastern neighbour (periodic
channel).  Run A moves the staging buffers with a device-to-device copy callback, run B with RCCL (communicator of
size 1: ncclSend / ncclRecv to self inside one group, enqueued on the kernels' stream).  Both with the per-point
exchanges and with the wide-halo external mode; all four results must be bit-identical."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dataclasses

import numpy as np
import torch

from extpom_amd import decomp
from extpom_amd.cases import make_case
from extpom_amd.halo import _DevPtr, rccl_library_path
from extpom_amd.layout import BLK2D, BLK3D
from extpom_amd.lib import OPP
from extpom_amd.model import PomGpu, gpu_finish_initial

IM, JM, KB, STEPS, ISPLIT = 128, 64, 12, 3, 6
SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}


def run(mover, wide):
    tile = decomp.make_tile(0, IM, JM, IM, JM, n_proc=1)
    tile = dataclasses.replace(tile, n_west=0, n_east=0)          # its own neighbour in x
    st = make_case("island", IM, JM, KB, dte=6.0, isplit=ISPLIT)
    gpu_finish_initial(st, device=0)
    st.n_west = st.n_east = 0
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    g = PomGpu(st, device=0, stream=ts.cuda_stream)
    dev = torch.device("cuda", 0)
    if mover == "copy":
        w = lambda p, n: torch.as_tensor(_DevPtr(p, (n,)), device=dev)

        def fn(send, scount, recv, rcount):
            with torch.cuda.stream(torch.cuda.ExternalStream(g.current_stream())):   # the stream of the round: the kernels' or the library's second one
                for d in range(8):
                    if tile_nb[d] >= 0 and rcount[d]:
                        w(recv[d], rcount[d]).copy_(w(send[OPP[d]], scount[OPP[d]]))
        tile_nb = PomGpu.neighbours8(tile)
        g.set_transport(tile, fn, agree=lambda mine: mine, stream_ordered=True)        # one rank: its own answer is the minimum; the copies are enqueued on the round's stream
    else:
        lib = rccl_library_path()
        g.rccl_init(tile, g.rccl_unique_id(lib), 0, 1, lib)
    if wide:
        assert g.set_wide_external(True, IM, JM)
    g.run(STEPS)
    g.download()
    n, ns = g.exchange_rounds(), g.exchange_rounds_side()
    out = {f: st.field(f).copy() for f in BLK2D + BLK3D if f not in SCRATCH}
    err = int(st.error_status)
    g.close()
    return out, n, err, ns


def main():
    ref, n_ref, err, _ = run("copy", False)
    assert err == 0 and n_ref > 100, (err, n_ref)
    assert np.isfinite(ref["u"]).all() and np.abs(ref["u"]).max() > 0
    for mover, wide, overlap, split_fails in (("rccl", False, True, False), ("copy", True, True, False), ("rccl", True, True, False), ("rccl", True, False, False),
                                              ("rccl", True, True, True)):
        # overlap: the early part of the wide exchange and the wr round on the library's second stream, over the second
        # (split) communicator, and the rim rounds -- nine of a step's ten rounds beside the kernels; POMGPU_NO_OVERLAP keeps one stream.
        # split_fails: the rank behaves as if ncclCommSplit had failed -- the ranks' agreement (ncclAllReduce(min) inside
        # pomgpu_rccl_init) then keeps every round on the main stream
        if overlap:
            os.environ.pop("POMGPU_NO_OVERLAP", None)
        else:
            os.environ["POMGPU_NO_OVERLAP"] = "1"
        if split_fails:
            os.environ["POMGPU_TEST_SPLIT_FAIL_RANK"] = "0"
        got, n, err, ns = run(mover, wide)
        os.environ.pop("POMGPU_NO_OVERLAP", None)
        os.environ.pop("POMGPU_TEST_SPLIT_FAIL_RANK", None)
        bad = [f for f in ref if not np.array_equal(ref[f], got[f])]
        assert err == 0 and not bad, (mover, wide, err, bad[:10])
        # on the second stream: every step the early part of the wide exchange, advct's edge lines, advx + advy + aam and wr; from the second
        # step on (the first skips mode_internal's 3-D body) w, the turbulence arrays, T / S / rho and the two velocity rounds that end
        # mode_internal (pomgpu_api.hip, "rim rounds")
        assert ns == (4 + 9 * (STEPS - 1) if wide and overlap and not split_fails else 0), (mover, wide, overlap, split_fails, ns)
        print(f"{mover} wide={wide} overlap={overlap} split_fails={split_fails}: {n} message rounds on the kernels' stream + {ns} on the side stream "
              f"(per-point, copy mover: {n_ref}), fields identical")
    print("RCCL-SELF-OK")


if __name__ == "__main__":
    main()
