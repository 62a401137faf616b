"""Kernel LOGIC on the CPU: the unmodified kernel sources of extpom_amd/csrc are compiled for the
host by tests/emu (each launch runs as a serial loop over its grid) and must reproduce the CPU
oracle bit for bit -- per routine and per step.  This is test infrastructure: the emulated
library lives in tests/_emu and is loaded only here (the product path needs a real HIP device).
The arrays tps, fluxua, fluxva, zflux are pure scratch in the reference;
the fused kernels do not materialise them, so they are not compared."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from extpom_amd.cases import make_case
from extpom_amd.layout import BLK2D, BLK3D
from extpom_amd.model import PomGpu
from oracle.pyoracle import OracleTile, oracle_finish_initial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "_emu", "libpomgpu_emu.so")
SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}


@pytest.fixture(scope="module", autouse=True)
def emu_lib():
    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)


def diff(a, b, skip=SCRATCH):
    return [n for n in BLK2D + BLK3D if n not in skip and not np.array_equal(a.field(n), b.field(n))]


@pytest.mark.parametrize("case,nml,steps", [
    ("seamount", dict(), 8), ("island", dict(), 6), ("basin", dict(), 6),
    ("seamount", dict(nadv=1), 4), ("island", dict(nitera=2), 4), ("seamount", dict(nitera=3, sw=1.0), 3),
    ("seamount", dict(mode=4), 4), ("seamount", dict(mode=2), 4), ("seamount", dict(nbct=3, nbcs=3), 4),
    ("basin", dict(isplit=10, alpha=0.225), 4), ("seamount", dict(isplit=7), 3),
    ("island", dict(npg=2), 4), ("seamount", dict(npg=2, nadv=1), 3)])   # odd isplit: the buffer generations end swapped
def test_steps_bit_identical(case, nml, steps):
    kw = dict(dte=6.0, isplit=30)
    kw.update(nml)
    a = make_case(case, 65, 49, 21, **kw)
    oracle_finish_initial(a)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b, libpath=EMU)
    for n in range(1, steps + 1):
        ot.run(1)
        g.run(1)
        g.download()
        assert not diff(a, b), f"step {n}: {diff(a, b)}"
        assert a.iint == b.iint == n
    assert g.check_velocity() == ot.vamax


@pytest.mark.parametrize("nbct,kb", [(2, 21), (4, 21), (2, 70)])
def test_short_wave_penetration_bit_identical(nbct, kb):
    """proft with nbc = 2 / 4 and swrad != 0: the reference evaluates the radiation term in REAL(16) and rounds once
    (solver.f:1608-1611; the oracle with libquadmath), the kernels in double-double arithmetic (csrc/dd_exp.h) -- the same double.
    kb = 70 takes the kernel with work vectors (k_proft), kb = 21 the register-resident one (k_proft_reg<24, 1>)."""
    a = make_case("seamount", 65, 49, kb, dte=6.0, isplit=30, nbct=nbct)
    a.swrad[...] = -5.0e-5 * a.fsm
    oracle_finish_initial(a)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b, libpath=EMU)
    for n in range(1, 7):
        ot.run(1)
        g.run(1)
        g.download()
        assert not diff(a, b), f"step {n}: {diff(a, b)}"


@pytest.mark.parametrize("case,im,jm", [("seamount", 66, 50), ("island", 128, 12)])
def test_even_leading_dimension_steps(case, im, jm):
    """an even im_local selects the two-columns-per-lane kernels (16-byte loads)"""
    a = make_case(case, im, jm, 21, dte=6.0, isplit=30)
    oracle_finish_initial(a)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b, libpath=EMU)
    for n in range(1, 4):
        ot.run(1)
        g.run(1)
        g.download()
        assert not diff(a, b), f"step {n}: {diff(a, b)}"


@pytest.mark.parametrize("canonical", [True, False])
def test_marching_external_substep_with_and_without_canonical_areas(monkeypatch, canonical):
    """k_ext_march (forced onto a small grid, ragged last segment): art, aru, arv formed from dx, dy where the arrays ARE those
    formulas (k_check_areas), read from memory where the host's arrays differ -- here aru and art perturbed in the last bit
    on a few cells: the results must follow the arrays, as the oracle's do"""
    monkeypatch.setenv("POMGPU_EXT_MARCH", "1")
    monkeypatch.setenv("POMGPU_EXT_ROWS", "6")
    a = make_case("seamount", 70, 45, 11, dte=6.0, isplit=10)
    oracle_finish_initial(a)
    if not canonical:
        for name in ("aru", "art", "arv"):
            f = a.field(name)
            f[5:30:3, 7:60:5] = np.nextafter(f[5:30:3, 7:60:5], np.inf)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b, libpath=EMU)
    for n in range(1, 3):
        ot.run(1)
        g.run(1)
        g.download()
        assert not diff(a, b), f"step {n}: {diff(a, b)}"


def test_surface_and_lateral_forcing_across_record_changes():
    """wind / heat / surface (bounds_forcing.f:871-983) and lateral_bc (:593-868) on the device side, called by
    advance itself (advance.f:14-18) once records are supplied; dti = 360 s puts the record changes of the surface
    fields at step 30 and those of the lateral boundary values at every tenth step"""
    from extpom_amd.cases import make_forcing_records, make_lateral_records
    a = make_case("seamount", 65, 49, 21, dte=6.0, isplit=60, days=1.0)
    oracle_finish_initial(a)
    make_forcing_records(a, 4)
    make_lateral_records(a, 6)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b, libpath=EMU)
    g.set_forcing_records()
    g.set_lateral_records()
    for n in range(1, 33):
        if n % 10 == 0:
            g.set_lateral_records(first=n // 10 + 2, count=1)      # the record lateral_bc asks for at this step
        ot.run(1)
        g.run(1)
        if n in (1, 2, 9, 10, 11, 20, 21, 29, 30, 31, 32):
            g.download()
            assert not diff(a, b) and np.array_equal(a.bdry, b.bdry), f"step {n}: {diff(a, b)}"


def test_output_and_restart_files_read_back(tmp_path):
    """the CDF-2 writer (host code of the library, io_pnetcdf.F:57-410, :1661-2083) through the emulated build:
    scipy's NetCDF reader sees the reference's dimensions, variables and attribute texts and the state's values"""
    from scipy.io import netcdf_file
    from extpom_amd.layout import RESTART_2D, RESTART_3D
    a = make_case("island", 65, 49, 21, dte=6.0, isplit=30)
    oracle_finish_initial(a)
    g = PomGpu(a, libpath=EMU)
    g.run(2)
    g.write_file("output", tmp_path / "out.nc", title="island", time_start="2000-01-01 00:00:00 +00:00")
    g.write_file("restart", tmp_path / "rst.nc", title="island", time_start="2000-01-01 00:00:00 +00:00")
    g.download()
    # the WHOLE header of both files against what the reference's own source defines (tests/golden/cdf_schema.json)
    from cdf_check import check_header
    for kind, name in (("output", "out.nc"), ("restart", "rst.nc")):
        check_header(tmp_path / name, kind, "island", "2000-01-01 00:00:00 +00:00", 21, 65, 49)
    with netcdf_file(str(tmp_path / "out.nc"), "r", mmap=False) as f:
        assert f.version_byte == 2 and f.description == b"output file"
        assert dict(f.dimensions) == dict(time=1, z=21, zz=20, y=49, x=65)
        assert list(f.variables)[:9] == ["time", "vtot", "mtot", "tavg", "savg", "eavg", "ekin", "z", "zz"] and list(f.variables)[-6:] == ["u", "v", "t", "s", "rho", "w"]
        assert f.variables["rho"].dimensions == ("time", "zz", "y", "x") and f.variables["rho"].units == b"dimensionless"
        assert np.array_equal(f.variables["t"][0], a.t[:20]) and np.array_equal(f.variables["w"][0], a.w) and np.array_equal(f.variables["elb"][0], a.elb)
    with netcdf_file(str(tmp_path / "rst.nc"), "r", mmap=False) as f:
        assert list(f.variables)[:2] == ["iint", "time"] and sorted(list(f.variables)[2:]) == sorted(RESTART_2D + RESTART_3D)
        for n in RESTART_2D + RESTART_3D:
            assert np.array_equal(f.variables[n][:], a.field(n)), n


def test_files_written_with_the_callers_statistics_see_the_current_state(tmp_path):
    """a caller that hands the writer its own (rank-reduced) statistics skips pomgpu_domain_stats -- and with it the side effect
    of bringing lazily kept arrays up to date.  After an ODD number of fused external substeps the current generation of
    ua, va, el, elb, uab, vab lives in the second buffer set: the file must hold it all the same (pomgpu_materialize)"""
    from scipy.io import netcdf_file
    a = make_case("seamount", 65, 49, 21, dte=6.0, isplit=7)
    oracle_finish_initial(a)
    g = PomGpu(a, libpath=EMU)
    g.run(3)
    g.write_file("output", tmp_path / "out.nc", title="t", time_start="s", stats=(1., 2., 3., 4., 5., 6., 7., 8.))
    g.write_file("restart", tmp_path / "rst.nc", title="t", time_start="s", stats=(1., 2., 3., 4., 5., 6., 7., 8.))
    g.download()
    with netcdf_file(str(tmp_path / "out.nc"), "r", mmap=False) as f:
        assert float(f.variables["vtot"][0]) == 1.0 and float(f.variables["ekin"][0]) == 8.0
        for n in ("uab", "vab", "elb"):
            assert np.array_equal(f.variables[n][0], a.field(n)), n
    with netcdf_file(str(tmp_path / "rst.nc"), "r", mmap=False) as f:
        for n in ("ua", "va", "el", "elb", "uab", "vab", "rho"):
            assert np.array_equal(f.variables[n][:], a.field(n)), n


def test_ramped_forcing_steps():
    """lramp = .true.: ramp = time/period changes every step (advance.f:66-72)"""
    a = make_case("seamount", 65, 49, 21, dte=6.0, isplit=30)
    a.lramp = True
    oracle_finish_initial(a)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b, libpath=EMU)
    ot.run(5)
    g.run(5)
    g.download()
    assert 0.0 < a.ramp < 1.0 and a.ramp == b.ramp and not diff(a, b), diff(a, b)


def test_kb_above_the_register_kernels_bound():
    """kb = 70 > 64: the column kernels with private work vectors take over from the unrolled ones"""
    a = make_case("basin", 64, 48, 70, dte=6.0, isplit=30)
    oracle_finish_initial(a)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b, libpath=EMU)
    for n in range(1, 3):
        ot.run(1)
        g.run(1)
        g.download()
        assert not diff(a, b), f"step {n}: {diff(a, b)}"


FALLBACK_ENV = ("POMGPU_THOMAS_SCRATCH", "POMGPU_NO_PAIR", "POMGPU_EXT_SPLIT", "POMGPU_ADVQ_SINGLE", "POMGPU_ADVT2_SINGLE",
                "POMGPU_REALVERTVL_CELLS", "POMGPU_BAROPG_CELLS", "POMGPU_VERTVL_CELLS")


@pytest.mark.parametrize("switches", [FALLBACK_ENV, ("POMGPU_ADVAVE_SEPARATE", "POMGPU_EXT_RIM_KERNEL"), ("POMGPU_PROFQ_ROWS8", "POMGPU_COL_STRIP", "POMGPU_EXT_MARCH"),
                                      ("POMGPU_PROFQ_ROWS8", "POMGPU_COL_STRIP", "POMGPU_EXT_MARCH", "POMGPU_NO_LIN")])
def test_general_kernels_behind_the_fast_paths(monkeypatch, switches):
    """the scratch-vector / one-column-per-lane / split kernels that serve kb > 64, odd im_local and
    multi-tile runs stay bit-identical too (selected here through the library's developer switches); second set:
    the external substep with advave and the rim cells as kernels of their own (ispadv != 1, mode = 2 take that path); third:
    the launch geometry of wide tiles -- k_profq's 8-row workgroups and the strip order of the row-sharing kernels (strips 3
    workgroups wide on a row of 4: a full and a ragged strip) -- every workgroup decoded exactly once, in the balanced XCD order
    this ragged grid takes by default and (fourth) in the banded one"""
    for v in switches:
        monkeypatch.setenv(v, "3" if v == "POMGPU_COL_STRIP" else "1")
    im, jm = (200, 30) if "POMGPU_COL_STRIP" in switches else (65, 49)
    a = make_case("seamount", im, jm, 21, dte=6.0, isplit=30)
    oracle_finish_initial(a)
    b = a.copy()
    ot = OracleTile(a)
    g = PomGpu(b, libpath=EMU)
    for n in range(1, 4):
        ot.run(1)
        g.run(1)
        g.download()
        assert not diff(a, b), f"step {n}: {diff(a, b)}"


def warm_state(case="island"):
    a = make_case(case, 65, 49, 21, dte=6.0, isplit=30)
    oracle_finish_initial(a)
    OracleTile(a).run(3)
    a.iint = 4
    a.iext = 7
    return a


ROUTINES = [
    ("advave", (), ()), ("advct", (), ()), ("advu", (), ()), ("advv", (), ()), ("baropg", (), ()), ("baropg_mcc", (), ()), ("profq", (), ()),
    ("profu", (), ()), ("profv", (), ()), ("vertvl", (), ()), ("realvertvl", (), ()), ("lateral_viscosity", (), ()),
    ("mode_interaction", (), ()), ("mode_external", (), ()), ("mode_internal", (), ()),
    ("advq", ("q2b", "q2", "uf"), ()), ("advt1", ("tb", "t", "tclim", "uf"), ()), ("advt2", ("sb", "s", "sclim", "vf"), ()),
    ("dens", ("s", "t", "rho"), ()), ("proft", ("uf", "wtsurf", "tsurf"), (1,)), ("proft", ("vf", "wssurf", "ssurf"), (3,)),
    ("bcond", (), (1,)), ("bcond", (), (2,)), ("bcond", (), (4,)), ("bcond", (), (5,)), ("bcond", (), (6,)),
    ("bcondorl", (), (3,)), ("bcondorl", (), (5,)), ("restore_interior", (), ()),
]


@pytest.mark.parametrize("name,fields,ints", ROUTINES, ids=[f"{r[0]}{''.join(map(str, r[2]))}" for r in ROUTINES])
def test_routine_bit_identical(name, fields, ints):
    a = warm_state()
    b = a.copy()
    ot = OracleTile(a)
    ot.call(name, *[ot.a3(f) for f in fields], *[ctypes.c_int(i) for i in ints])
    g = PomGpu(b, libpath=EMU)
    g.call(name, *fields, *ints)
    g.download()
    # stand-alone profu/profv leave the reference's non-interior junk of advu/advv untouched: same input, same output
    assert not diff(a, b), f"{name}: {diff(a, b)}"


def test_trimmed_tile_padding_is_inert():
    """a tile whose active extent is smaller than its leading dimensions (east/north-most tiles,
    parallel_mpi.f:83-87) gives the same active-region result as the tight layout"""
    from extpom_amd.layout import PomState
    a = make_case("basin", 65, 49, 21, dte=6.0, isplit=30)
    oracle_finish_initial(a)
    b = PomState(72, 53, 21, im=65, jm=49)
    b.blk1d[...] = a.blk1d
    b.blk2d[:, :49, :65] = a.blk2d
    b.blk3d[:, :, :49, :65] = a.blk3d
    from extpom_amd.layout import BDRY
    for n, kind in BDRY:
        src, dst = a.field(n), b.field(n)
        if kind == "J":
            dst[:49] = src
        elif kind == "I":
            dst[:65] = src
        elif kind == "JK":
            dst[:, :49] = src
        else:
            dst[:, :65] = src
    b.con[...] = a.con
    b.restore_records = a.restore_records
    ga, gb = PomGpu(a, libpath=EMU), PomGpu(b, libpath=EMU)
    ga.run(3)
    gb.run(3)
    ga.download()
    gb.download()
    for n in BLK2D:
        if n not in SCRATCH:
            assert np.array_equal(a.field(n), b.field(n)[:49, :65]), n
    for n in BLK3D:
        if n not in SCRATCH:
            assert np.array_equal(a.field(n), b.field(n)[:, :49, :65]), n
