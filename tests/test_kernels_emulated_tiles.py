"""The MULTI-TILE kernel paths on the CPU: the host build of the kernel sources (tests/emu) runs one tile
per Python thread; every exchange point of the C ABI hook is served by a barrier-synchronised copy between
the tiles' (host-resident) "device" arrays with the semantics of exchange2d_mpi / exchange3d_mpi and
order2d_mpi / order3d_mpi.  The owned cells of all tiles must equal the single-tile CPU oracle bit for bit.
This covers, without a GPU, the split kernels that only run when a context has an exchange hook (advct
a/b/c, advave a/b/c, the three external-mode kernels, advq flux/step, profq with its own production
kernel) and baropg_mcc's extra ghost column / row."""
import ctypes
import os
import subprocess
import threading

import numpy as np
import pytest

from extpom_amd import decomp
from extpom_amd.cases import finish_initial, make_case
from extpom_amd.layout import BLK2D, BLK3D
from extpom_amd.model import PomGpu
from oracle.pyoracle import OracleTile, oracle_finish_initial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "_emu", "libpomgpu_emu.so")
SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}
IM, JM, KB, STEPS = 41, 35, 11, 3


@pytest.fixture(scope="module", autouse=True)
def emu_lib():
    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)


class Board:
    """what the ranks of one run share: a mailbox per (sender, receiver) and a barrier"""

    def __init__(self, world):
        self.box = {}
        self.barrier = threading.Barrier(world)


def view(ptr, nz, tile):
    n = nz * tile.jm_local * tile.im_local
    return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr)).reshape(nz, tile.jm_local, tile.im_local)


def exchange(board, tile, arrays):
    """parallel_mpi.f:154-351: east/west first, then north/south including the fresh ghost columns"""
    im, jm, me = tile.im, tile.jm, tile.rank
    for lo, hi, take, put in (
            (tile.n_west, tile.n_east, lambda a, s: a[:, :jm, im - 2 if s == "hi" else 1].copy(),
             lambda a, s, v: a[:, :jm, im - 1 if s == "hi" else 0].__setitem__(slice(None), v)),
            (tile.n_south, tile.n_north, lambda a, s: a[:, jm - 2 if s == "hi" else 1, :im].copy(),
             lambda a, s, v: a[:, jm - 1 if s == "hi" else 0, :im].__setitem__(slice(None), v))):
        for nb, side in ((hi, "hi"), (lo, "lo")):
            if nb >= 0:
                board.box[(me, nb)] = [take(a, side) for a in arrays]
        board.barrier.wait()
        for nb, side in ((hi, "hi"), (lo, "lo")):
            if nb >= 0:
                for a, v in zip(arrays, board.box[(nb, me)]):
                    put(a, side, v)
        board.barrier.wait()


OPPOSITE = dict(w="e", e="w", s="n", n="s", sw="ne", ne="sw", se="nw", nw="se")
DIRS = ("w", "e", "s", "n", "sw", "se", "nw", "ne")


def exchange8(board, tile, g, ptrs, nzs):
    """the library's single-round exchange (pomgpu_halo_pack8 / unpack8) with up to eight neighbours"""
    me = tile.rank
    nb = dict(w=tile.n_west, e=tile.n_east, s=tile.n_south, n=tile.n_north, sw=tile.n_sw, se=tile.n_se, nw=tile.n_nw, ne=tile.n_ne)
    length = dict(w=tile.jm, e=tile.jm, s=tile.im, n=tile.im, sw=1, se=1, nw=1, ne=1)
    total = sum(nzs)
    count = len(ptrs)
    cp = (ctypes.c_void_p * count)(*ptrs)
    cn = (ctypes.c_int * count)(*nzs)
    send = {d: np.zeros(total * length[d]) for d in DIRS if nb[d] >= 0}
    recv = {d: np.zeros(total * length[d]) for d in DIRS if nb[d] >= 0}
    tab = lambda bufs: (ctypes.c_void_p * 8)(*[bufs[d].ctypes.data if d in bufs else None for d in DIRS])
    assert g.L.pomgpu_halo_pack8(g.h, cp, cn, count, tab(send)) == 0
    for d in send:
        board.box[(me, nb[d], d)] = send[d]
    board.barrier.wait()
    for d in recv:                                    # what arrives from direction d left the neighbour towards OPPOSITE[d]
        recv[d][:] = board.box[(nb[d], me, OPPOSITE[d])]
    assert g.L.pomgpu_halo_unpack8(g.h, cp, cn, count, tab(recv)) == 0
    board.barrier.wait()


def order(board, tile, send_e, n_e, send_n, n_n, recv_w, recv_s):
    """parallel_mpi.f:353-480: one-way, eastward and northward"""
    me = tile.rank
    buf = lambda p, n: np.ctypeslib.as_array((ctypes.c_double * n).from_address(p))
    if tile.n_east >= 0:
        board.box[(me, tile.n_east, "o")] = buf(send_e, n_e).copy()
    if tile.n_north >= 0:
        board.box[(me, tile.n_north, "o")] = buf(send_n, n_n).copy()
    board.barrier.wait()
    if tile.n_west >= 0:
        buf(recv_w, n_e)[:] = board.box[(tile.n_west, me, "o")]
    if tile.n_south >= 0:
        buf(recv_s, n_n)[:] = board.box[(tile.n_south, me, "o")]
    board.barrier.wait()


def run_tiles(nx, ny, nml, single_round=False):
    world = nx * ny
    iml, jml = decomp.local_size(IM, JM, nx, ny)
    board, out, errs = Board(world), {}, []

    def rank(r):
        try:
            tile = decomp.make_tile(r, IM, JM, iml, jml, n_proc=world)
            st = make_case("island", IM, JM, KB, tile=tile, dte=6.0, isplit=10, **nml)
            g = PomGpu(st, libpath=EMU)
            count = [0]

            def hook(ptrs, nzs):
                count[0] += 1
                if single_round:
                    exchange8(board, tile, g, ptrs, nzs)
                else:
                    exchange(board, tile, [view(p, nz, tile) for p, nz in zip(ptrs, nzs)])

            g.set_exchange(hook)
            g.set_order_exchange(lambda *a: order(board, tile, *a))

            def dens(s, a, b, c):
                g.upload(s); g.call("dens", a, b, c); g.download(s)

            def baropg(s):
                g.upload(s); g.call("baropg_mcc" if int(s.npg) == 2 else "baropg"); g.download(s)

            finish_initial(st, dens, baropg)
            g.upload(st)
            g.run(STEPS)
            g.download()
            out[r] = (tile, st, count[0])
        except Exception as e:                      # a dead rank must not leave the others at the barrier
            errs.append(e)
            board.barrier.abort()

    threads = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    return out


@pytest.mark.parametrize("nx,ny,nml,single_round", [(2, 1, {}, False), (1, 2, {}, False), (2, 2, {}, False), (2, 2, dict(npg=2), False),
                                                    (2, 1, dict(nadv=1, mode=3), False),
                                                    (2, 2, {}, True), (3, 2, dict(npg=2), True), (1, 3, {}, True)])
def test_tiles_match_single_tile_oracle(nx, ny, nml, single_round):
    """single_round: the library's one-round exchange with eight neighbours instead of the reference's two phases"""
    out = run_tiles(nx, ny, nml, single_round)
    g = make_case("island", IM, JM, KB, dte=6.0, isplit=10, **nml)
    oracle_finish_initial(g)
    OracleTile(g).run(STEPS)
    bad = []
    for r, (tile, st, count) in out.items():
        assert count > 100
        io, jo, im, jm = tile.i_off, tile.j_off, tile.im, tile.jm
        sl_j = slice(0 if jo == 0 else 1, jm if jo + jm == JM else jm - 1)       # the cells the tile owns
        sl_i = slice(0 if io == 0 else 1, im if io + im == IM else im - 1)
        for n in BLK2D + BLK3D:
            if n in SCRATCH:
                continue
            ref = g.field(n)[..., jo:jo + jm, io:io + im][..., sl_j, sl_i]
            got = st.field(n)[..., :jm, :im][..., sl_j, sl_i]
            if not np.array_equal(ref, got):
                bad.append((r, n))
    assert not bad, bad
