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


@pytest.fixture(params=["side stream at once", "side stream as late as possible"], autouse=True)
def side_stream_timing(request, monkeypatch):
    """Every test of this file runs twice.  The host build executes a launch where it is enqueued, i.e. work on the library's second
    stream at the EARLIEST moment the device could run it; with POMGPU_EMU_DEFER_SIDE the emulated runtime (tests/emu/hip/hip_runtime.h)
    keeps that stream's launches in a queue and runs them at the LATEST moment -- when the main stream waits for an event behind them, or
    when the host needs the stream (a message round of a callback mover).  A main-stream kernel that does not wait for the side-stream
    round it depends on, or side-stream work whose operands the main stream overwrites before the join, gives other bits in one of the
    two.  (What neither shows: a missing wait ACROSS the next side-stream round, because the callback mover completes the stream when
    a round is posted; tests/gpu_rccl_self.py and the asynchronous mover of tests/gpu_tiles_threads.py cover that on the GPU.)"""
    if request.param.endswith("possible"):
        monkeypatch.setenv("POMGPU_EMU_DEFER_SIDE", "1")
    else:
        monkeypatch.delenv("POMGPU_EMU_DEFER_SIDE", raising=False)


class Board:
    """what the ranks of one run share: a mailbox per (sender, receiver) and a barrier"""

    def __init__(self, world):
        self.box = {}
        self.barrier = threading.Barrier(world)

    def allmin(self, me, value):
        """the host's reduction behind pomgpu_transport_side_agree: the minimum of `value` over all ranks"""
        self.box[("min", me)] = int(value)
        self.barrier.wait()
        m = min(v for k, v in self.box.items() if k[0] == "min")
        self.barrier.wait()
        return m


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


def transport(board, tile, send, scount, recv, rcount):
    """the mover behind pomgpu_set_transport: what leaves towards direction d arrives at the neighbour from OPP[d]"""
    me = tile.rank
    nb = PomGpu.neighbours8(tile)
    buf = lambda p, n: np.ctypeslib.as_array((ctypes.c_double * n).from_address(p))
    for d in range(8):
        if nb[d] >= 0 and scount[d]:
            board.box[(me, nb[d], d)] = buf(send[d], scount[d]).copy()
    board.barrier.wait()
    for d in range(8):
        if nb[d] >= 0 and rcount[d]:
            msg = board.box[(nb[d], me, OPP8[d])]
            assert msg.size == rcount[d], (me, d, msg.size, rcount[d])
            buf(recv[d], rcount[d])[:] = msg
    board.barrier.wait()


OPP8 = (1, 0, 3, 2, 7, 6, 5, 4)


def side_rounds(steps, nml=None):
    """message rounds the library serves on its second stream in `steps` internal steps from a cold start (pomgpu_api.hip, "rim rounds"): every
    step the early part of the wide exchange, advct's edge lines (R1), advx + advy + aam (R2) and wr; every step but the first (which skips
    mode_internal's 3-D body, advance.f:362) also w (Rw), the turbulence arrays (Rq), T / S / rho (Rts: only behind the one-pass tracer
    advection, nadv = 2 with nitera = 1) and the two velocity rounds that end mode_internal (R7, R8)"""
    nml = nml or {}
    rts = nml.get("nadv", 2) == 2 and nml.get("nitera", 1) == 1 and nml.get("mode", 3) != 4
    return 4 + (8 + (1 if rts else 0)) * (steps - 1)


def main_rounds_saved(steps):
    """how many rounds fewer the kernels' own stream carries for it (default namelist): all of a full step's ten but the late part of the wide exchange"""
    return 3 + 8 * (steps - 1)


def run_tiles(nx, ny, nml, single_round=False, library_exchange=False, wide=False, grid=None, isplit=10, case="island", steps=None,
              by_routine=False, records=False, dte=6.0, side_fail_rank=None, side_rounds=None, rank_switches=None, errors=None):
    world = nx * ny
    IMg, JMg = grid or (IM, JM)
    iml, jml = decomp.local_size(IMg, JMg, nx, ny)
    board, out, errs = Board(world), {}, []
    tiles = [decomp.make_tile(r, IMg, JMg, iml, jml, n_proc=world) for r in range(world)]

    def rank(r):
        try:
            tile = tiles[r]
            st = make_case(case, IMg, JMg, KB, tile=tile, dte=dte, isplit=isplit, **nml)
            g = PomGpu(st, libpath=EMU)
            for k, v in (rank_switches or {}).get(r, {}).items():   # as if this rank alone had been started with POMGPU_<k>=v
                g.switch(k, v)
            count = [0]

            def hook(ptrs, nzs):
                count[0] += 1
                if single_round:
                    exchange8(board, tile, g, ptrs, nzs)
                else:
                    exchange(board, tile, [view(p, nz, tile) for p, nz in zip(ptrs, nzs)])

            if library_exchange:                     # the library packs, moves and unpacks by itself
                # side-stream rounds: the ranks' own answers reduced to their minimum (side_fail_rank: that rank says no)
                g.set_transport(tile, lambda *a: transport(board, tile, *a),
                                agree=lambda mine: board.allmin(r, 0 if (r == side_fail_rank and mine in (1, 2)) else mine))   # 1, 2: the capability answers (pomgpu.h); the switch digest passes unchanged
                if wide:
                    assert g.set_wide_external(True, min(t.im for t in tiles), min(t.jm for t in tiles))
            else:
                g.set_exchange(hook)
                g.set_order_exchange(lambda *a: order(board, tile, *a))

            def dens(s, a, b, c):
                g.upload(s); g.call("dens", a, b, c); g.download(s)

            def baropg(s):
                g.upload(s); g.call("baropg_mcc" if int(s.npg) == 2 else "baropg"); g.download(s)

            finish_initial(st, dens, baropg)
            if records:                              # file forcing: wind / heat / surface and lateral_bc records (cases.py)
                from extpom_amd.cases import make_forcing_records, make_lateral_records
                make_forcing_records(st, 4)
                make_lateral_records(st, 6)
            g.upload(st)
            if records:
                g.set_forcing_records()
                g.set_lateral_records()
                for n in range(1, (steps or STEPS) + 1):
                    if n % records == 0:
                        g.set_lateral_records(first=n // records + 2, count=1)   # the record lateral_bc asks for at this step
                    g.run(1)
            elif by_routine:                           # the reference's own sequence (advance.f:6-59), one entry point per subroutine,
                for n in range(1, (steps or STEPS) + 1):   # as the Fortran host drives it
                    g.set_con(iint=n)
                    g.call("get_time")
                    g.get_con()                      # time, ramp: the host's copy of blkcon is what set_con pushes below
                    g.call("lateral_viscosity")
                    g.call("mode_interaction")
                    for iext in range(1, isplit + 1):
                        g.set_con(iext=iext)
                        g.call("mode_external")
                    g.set_con(iext=isplit + 1)
                    g.call("mode_internal")
                    g.check_velocity()
            else:
                g.run(steps or STEPS)
            g.download()
            out[r] = (tile, st, g.exchange_rounds() if library_exchange else count[0])
            if side_rounds is not None and library_exchange:
                side_rounds[r] = g.exchange_rounds_side()
        except Exception as e:                      # a dead rank must not leave the others at the barrier
            errs.append(e)
            if "different POMGPU_* switch sets" not in str(e):    # that refusal every rank reaches by itself: nobody is left waiting
                board.barrier.abort()

    threads = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors is not None:
        errors.extend(errs)
        return out
    assert not errs, errs
    return out


@pytest.mark.parametrize("nx,ny,nml,single_round", [(2, 1, {}, False), (1, 2, {}, False), (2, 2, {}, False), (2, 2, dict(npg=2), False),
                                                    (2, 1, dict(nadv=1, mode=3), False),
                                                    (2, 2, {}, True), (3, 2, dict(npg=2), True), (1, 3, {}, True)])
def test_tiles_match_single_tile_oracle(nx, ny, nml, single_round):
    """single_round: the library's one-round exchange with eight neighbours instead of the reference's two phases"""
    out = run_tiles(nx, ny, nml, single_round)
    compare_with_single_tile(out, nml)


def compare_with_single_tile(out, nml, grid=None, isplit=10, case="island", steps=None, min_rounds=20, ghosts=False):
    IMg, JMg = grid or (IM, JM)
    g = make_case(case, IMg, JMg, KB, dte=6.0, isplit=isplit, **nml)
    oracle_finish_initial(g)
    OracleTile(g).run(steps or STEPS)
    bad = []
    for r, (tile, st, count) in out.items():
        assert count > min_rounds, count
        io, jo, im, jm = tile.i_off, tile.j_off, tile.im, tile.jm
        sl_j = slice(0 if jo == 0 else 1, jm if jo + jm == JMg else jm - 1)       # the cells the tile owns
        sl_i = slice(0 if io == 0 else 1, im if io + im == IMg else im - 1)
        if ghosts:
            sl_i = sl_j = slice(None)
        for n in BLK2D + BLK3D:
            if n in SCRATCH:
                continue
            ref = g.field(n)[..., jo:jo + jm, io:io + im][..., sl_j, sl_i]
            got = st.field(n)[..., :jm, :im][..., sl_j, sl_i]
            if not np.array_equal(ref, got):
                bad.append((r, n))
    assert not bad, bad
    return {r: v[2] for r, v in out.items()}


@pytest.mark.parametrize("nx,ny,nml", [(2, 2, {}), (3, 2, dict(npg=2)), (1, 2, dict(nadv=1))])
def test_library_transport_serves_every_exchange_point(nx, ny, nml):
    """pomgpu_set_transport: pack8 / message round / unpack8 inside the library, baropg_mcc's order messages too"""
    out = run_tiles(nx, ny, nml, library_exchange=True)
    compare_with_single_tile(out, nml)


WIDE_GRID, WIDE_ISPLIT = (67, 59), 7        # w = 7 + 4 = 11 extra cells; 2x2 tiles of ~35 x 31


@pytest.mark.parametrize("nx,ny,case,nml", [(2, 2, "island", {}), (2, 1, "seamount", {}), (1, 2, "seamount", dict(npg=2)), (3, 2, "island", dict(nadv=1)),
                                            (1, 3, "seamount", {})])
def test_wide_halo_external_mode(nx, ny, case, nml):
    """pomgpu_set_wide_external: ONE wide exchange per internal step instead of six narrow ones per external substep;
    owned cells equal the single-tile oracle bit for bit, with a fraction of the message rounds.  (1 x 3: whole-row tiles, the
    middle one extended on both sides -- the rows its substeps skip as they go stale, row_window in pomgpu_api.hip, on either side)"""
    if nx == 3:
        grid = (97, 59)
    elif ny == 3:
        grid = (41, 101)
    else:
        grid = WIDE_GRID
    narrow = run_tiles(nx, ny, nml, library_exchange=True, grid=grid, isplit=WIDE_ISPLIT, case=case)
    n_narrow = compare_with_single_tile(narrow, nml, grid=grid, isplit=WIDE_ISPLIT, case=case, min_rounds=20)
    wide = run_tiles(nx, ny, nml, library_exchange=True, wide=True, grid=grid, isplit=WIDE_ISPLIT, case=case)
    n_wide = compare_with_single_tile(wide, nml, grid=grid, isplit=WIDE_ISPLIT, case=case, min_rounds=2)
    assert n_wide[0] < n_narrow[0] - 6 * WIDE_ISPLIT * STEPS + 3 * STEPS + 8, (n_wide, n_narrow)
    # ghost cells too: the tile's arrays are what the per-point exchanges leave there
    for r in wide:
        for n in ("ua", "va", "el", "elb", "d", "uab", "vab", "etf", "egf", "utf", "vtf", "adx2d", "ady2d", "advua", "advva", "elf", "uaf", "vaf"):
            t = wide[r][0]
            assert np.array_equal(wide[r][1].field(n)[:t.jm, :t.im], narrow[r][1].field(n)[:t.jm, :t.im]), (r, n)


@pytest.mark.parametrize("nx,ny,grid,case,nml", [(2, 4, (43, 75), "island", {}), (2, 4, (43, 75), "seamount", dict(npg=2)), (3, 3, (59, 53), "seamount", {}),
                                                 (3, 3, (59, 53), "island", dict(nadv=1))])
def test_baselines_own_2x4_split_and_a_tile_with_eight_neighbours(nx, ny, grid, case, nml):
    """BASELINE configs[2] / [3] name a 2 x 4 tile decomposition: (43, 75) splits into tiles of 23 x 21 whose north row is TRIMMED
    (jm = 18 of jm_local = 21, parallel_mpi.f:96-103) and every tile has a diagonal neighbour; 3 x 3 has a centre tile with all EIGHT
    neighbours live (parallel_mpi.f:111-119 reaches the diagonal ones through its two phases, the library sends them their corner
    cells, and the extended tile of the wide-halo mode takes a (w+1) x (w+1) corner block from each).  The library exchange and
    the wide-halo mode with its second-stream rounds against the single-tile oracle, owned cells bit for bit; and the two paths
    against each other on every cell, ghost cells included."""
    lib = run_tiles(nx, ny, nml, library_exchange=True, grid=grid, isplit=WIDE_ISPLIT, case=case)
    compare_with_single_tile(lib, nml, grid=grid, isplit=WIDE_ISPLIT, case=case, min_rounds=20)
    side = {}
    wide = run_tiles(nx, ny, nml, library_exchange=True, wide=True, grid=grid, isplit=WIDE_ISPLIT, case=case, side_rounds=side)
    compare_with_single_tile(wide, nml, grid=grid, isplit=WIDE_ISPLIT, case=case, min_rounds=2)
    assert set(side.values()) == {side_rounds(STEPS, nml)}, side
    tiles = [wide[r][0] for r in sorted(wide)]
    if ny == 4:
        assert {t.jm for t in tiles if t.py == 3} == {tiles[0].jm_local - 3} and all(t.jm == t.jm_local for t in tiles if t.py < 3)
    else:
        assert min(PomGpu.neighbours8(tiles[4])) >= 0                  # the centre tile: eight live neighbours
    bad = []
    for r in lib:
        t = lib[r][0]
        for n in BLK2D + BLK3D:
            if n not in SCRATCH and not np.array_equal(lib[r][1].field(n)[..., :t.jm, :t.im], wide[r][1].field(n)[..., :t.jm, :t.im]):
                bad.append((r, n))
    assert not bad, bad[:10]


def test_side_stream_rounds_are_a_collective_decision():
    """Rounds on the library's second stream (the early part of the wide exchange, wr) run on all ranks or on none: one rank of
    2x2 that reports it cannot serve them (a failed ncclCommSplit / hipStreamCreate in production) keeps EVERY rank on the
    main stream -- no rank posts a round its neighbours do not expect --, the results stay bit-identical, and the step
    has one round less per step (the wide exchange in one piece).  With every rank able, nine of a full step's ten rounds run on the
    second stream (side_rounds above)."""
    counts = {}
    for fail in (None, 2):
        side = {}
        out = run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT, side_fail_rank=fail, side_rounds=side)
        compare_with_single_tile(out, {}, grid=WIDE_GRID, isplit=WIDE_ISPLIT, min_rounds=3, ghosts=False)
        counts[fail] = ({r: v[2] for r, v in out.items()}, side)
    (main_all, side_all), (main_one, side_one) = counts[None], counts[2]
    assert set(side_one.values()) == {0}, side_one                     # nobody went to the second stream
    assert set(side_all.values()) == {side_rounds(STEPS)}, side_all    # early gather, R1, R2, wr every step, Rw, Rq, Rts, R7, R8 from the second step on: every rank
    for r in main_all:       # per full step: 1 round between kernels + 9 beside them, or 9 between kernels (the gather one round instead of early + late)
        assert main_one[r] == main_all[r] + main_rounds_saved(STEPS), (main_one, main_all, side_all)


def test_ranks_with_different_switch_sets_are_refused_together():
    """One rank of 2x2 was started with POMGPU_NO_SIDE_COMM (or any other switch that chooses which message rounds exist and
    where they run): the ranks compare a digest of those switches before anything collective depends on them and ALL of them
    refuse with a message -- nobody enters a round its neighbours will not post (transport.hip does the same over the RCCL
    communicator before ncclCommSplit).  A switch that only picks a kernel shape may differ from rank to rank."""
    from extpom_amd.lib import PomGpuError
    errs = []
    run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT, rank_switches={1: {"POMGPU_NO_SIDE_COMM": "1"}}, errors=errs)
    refused = [e for e in errs if isinstance(e, PomGpuError) and "different POMGPU_* switch sets" in str(e)]
    assert len(refused) == 4, errs                                     # every rank, by itself, before the first step
    side = {}
    out = run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT, rank_switches={1: {"PROFQ_ROWS2": "1", "NO_LIN": "1"}},
                    side_rounds=side)
    compare_with_single_tile(out, {}, grid=WIDE_GRID, isplit=WIDE_ISPLIT, min_rounds=2, ghosts=False)
    assert set(side.values()) == {side_rounds(STEPS)}, side
    # the same switch on EVERY rank is an agreement, not a difference: all of them keep their rounds on the main stream
    side = {}
    out = run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT,
                    rank_switches={r: {"NO_SIDE_COMM": "1"} for r in range(4)}, side_rounds=side)
    compare_with_single_tile(out, {}, grid=WIDE_GRID, isplit=WIDE_ISPLIT, min_rounds=2, ghosts=False)
    assert set(side.values()) == {0}, side


@pytest.mark.parametrize("switch,per_full_step,first_step", [("RIM_RESULTS_MAIN", 6, 4), ("RIM_MAIN", 2, 2)])
def test_rim_rounds_can_be_kept_on_the_kernels_stream(switch, per_full_step, first_step):
    """POMGPU_RIM_RESULTS_MAIN keeps the reference's three input exchanges of profq / proft (between kernels) instead of the result rounds
    Rw, Rq, Rts; POMGPU_RIM_MAIN keeps every rim round on the kernels' stream (the early part of the wide exchange and wr stay beside
    them): the same bits on every cell, ghost cells included, as the default (nine rounds of ten on the second stream) -- where a
    round runs and whether inputs or results travel changes nothing but the schedule.  Both switches belong to the collective digest."""
    side_a, side_b = {}, {}
    a = run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT, case="seamount", side_rounds=side_a)
    b = run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT, case="seamount", side_rounds=side_b,
                  rank_switches={r: {switch: "1"} for r in range(4)})
    assert set(side_a.values()) == {side_rounds(STEPS)} and set(side_b.values()) == {first_step + per_full_step * (STEPS - 1)}, (side_a, side_b)
    bad = []
    for r in a:
        t = a[r][0]
        for n in BLK2D + BLK3D:
            if n not in SCRATCH and not np.array_equal(a[r][1].field(n)[..., :t.jm, :t.im], b[r][1].field(n)[..., :t.jm, :t.im]):
                bad.append((r, n))
    assert not bad, bad[:10]
    for r in a:                                                        # the total number of rounds does not change
        assert a[r][2] + side_a[r] == b[r][2] + side_b[r], (r, a[r][2], side_a[r], b[r][2], side_b[r])


def test_wide_halo_too_narrow_shows_up(monkeypatch):
    """the stale rim of the extended tile grows by one cell per substep: with fewer than isplit - 1 extra cells it
    reaches owned cells and the comparison with the single-tile oracle must fail"""
    monkeypatch.setenv("POMGPU_WIDE_W", str(WIDE_ISPLIT - 2))
    out = run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT)
    with pytest.raises(AssertionError):
        compare_with_single_tile(out, {}, grid=WIDE_GRID, isplit=WIDE_ISPLIT, min_rounds=2)


@pytest.mark.parametrize("nx,ny,case,nml", [(3, 2, "seamount", {}), (2, 2, "island", dict(npg=2))])
def test_library_paths_leave_the_same_ghost_cells_as_the_reference_exchanges(nx, ny, case, nml):
    """The tile shortcuts of the library's own exchange (advct through edge lines, advq without its redundant flux
    exchange, profq's production term exchanged on the rim lines only, merged rounds, wide-halo external mode) against
    the hook path that keeps every exchange point of the reference: EVERY cell of every COMMON array, ghost cells
    included, must be identical -- what a download, an output file or a restart sees does not depend on the path."""
    grid = (97, 59)
    hooks = run_tiles(nx, ny, nml, grid=grid, isplit=WIDE_ISPLIT, case=case)
    for wide in (False, True):
        lib = run_tiles(nx, ny, nml, library_exchange=True, wide=wide, grid=grid, isplit=WIDE_ISPLIT, case=case)
        bad = []
        for r in hooks:
            t = hooks[r][0]
            for n in BLK2D + BLK3D:
                if n in SCRATCH:
                    continue
                if not np.array_equal(hooks[r][1].field(n)[..., :t.jm, :t.im], lib[r][1].field(n)[..., :t.jm, :t.im]):
                    bad.append((r, n))
        assert not bad, (wide, bad[:10])


def test_wide_halo_mode_under_the_reference_call_sequence():
    """the Fortran host calls get_time, lateral_viscosity, mode_interaction, isplit x mode_external, mode_internal,
    check_velocity one by one (advance.f:6-59): the wide-halo mode starts in mode_interaction and ends with the last
    mode_external, same results, same few message rounds"""
    out = run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT, by_routine=True)
    rounds = compare_with_single_tile(out, {}, grid=WIDE_GRID, isplit=WIDE_ISPLIT, min_rounds=2)
    assert rounds[0] < 12 * STEPS, rounds


def test_wide_halo_mode_with_file_forcing_records():
    """surface_forcing and lateral_bc (advance.f:14-18) change wusurf, wvsurf, e_atmos ... and the open-boundary lines every
    step: the wide exchange must carry them (and the boundary lines along the tile edges) -- 2x2 tiles of the seamount
    case (all four sides open) across a lateral record change, against the single-tile oracle with the same records"""
    from extpom_amd.cases import make_forcing_records, make_lateral_records
    grid, isplit, dte, steps = WIDE_GRID, 20, 6.0, 32          # dti = 120 s: lateral records (1/24 d) change at step 30; w = 24
    out = run_tiles(2, 2, {}, library_exchange=True, wide=True, grid=grid, isplit=isplit, case="seamount", steps=steps, records=30, dte=dte)
    g = make_case("seamount", grid[0], grid[1], KB, dte=dte, isplit=isplit)
    oracle_finish_initial(g)
    make_forcing_records(g, 4)
    make_lateral_records(g, 6)
    ot = OracleTile(g)
    ot.run(steps)
    assert np.isfinite(g.field("u")).all() and int(g.error_status) == 0
    bad = []
    for r, (tile, st, count) in out.items():
        io, jo, im, jm = tile.i_off, tile.j_off, tile.im, tile.jm
        sl_j = slice(0 if jo == 0 else 1, jm if jo + jm == grid[1] else jm - 1)
        sl_i = slice(0 if io == 0 else 1, im if io + im == grid[0] else im - 1)
        for n in BLK2D + BLK3D:
            if n in SCRATCH:
                continue
            if not np.array_equal(g.field(n)[..., jo:jo + jm, io:io + im][..., sl_j, sl_i], st.field(n)[..., :jm, :im][..., sl_j, sl_i]):
                bad.append((r, n))
    assert not bad, bad[:12]


def test_reference_shaped_kernels_under_the_library_exchange(monkeypatch):
    """the tile shortcuts switched off one level down: advct as its three kernels with whole-array exchanges, advq
    with its flux exchange, the production term as a full-array kernel -- still through pomgpu_set_transport"""
    for v in ("POMGPU_ADVCT_SPLIT", "POMGPU_ADVQ_EXCHANGE", "POMGPU_PROD_FULL"):
        monkeypatch.setenv(v, "1")
    out = run_tiles(2, 2, {}, library_exchange=True)
    compare_with_single_tile(out, {})


@pytest.mark.parametrize("nml", [dict(mode=2), dict(mode=4), dict(nitera=2)])
def test_wide_halo_mode_in_the_other_modes(nml):
    """mode = 2 (2-D only: advave's bottom-stress / curvature branch runs on the extended tile), mode = 4 (no tracer
    step), two Smolarkiewicz iterations (the general advt2 path with its own exchanges) -- 2x2 seamount tiles"""
    out = run_tiles(2, 2, nml, library_exchange=True, wide=True, grid=WIDE_GRID, isplit=WIDE_ISPLIT, case="seamount")
    compare_with_single_tile(out, nml, grid=WIDE_GRID, isplit=WIDE_ISPLIT, case="seamount", min_rounds=1)
