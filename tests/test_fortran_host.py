"""The Fortran host (extpom_amd/fortran): reference-named wrappers over the C ABI, driven by a
`program pom`-shaped main.  Build check runs everywhere flang exists; the run needs a GPU."""
import os
import subprocess

import numpy as np
import pytest

from extpom_amd.cases import make_case
from extpom_amd.layout import BLK2D, BLK3D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "extpom_amd", "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"
SCRATCH = {"tps", "fluxua", "fluxva", "zflux"}

needs_flang = pytest.mark.skipif(not os.path.exists(FLANG), reason="AMD flang not installed")


def _build():
    import __graft_entry__ as ge
    ge.build_hip()
    subprocess.check_call(["make", "-C", FDIR, "IM=65", "JM=49", "KB=21"], stdout=subprocess.DEVNULL)


@needs_flang
def test_fortran_host_builds_and_common_blocks_have_reference_sizes():
    _build()
    out = subprocess.run(["nm", "-S", os.path.join(FDIR, "pom_gpu_host.o")], capture_output=True, text=True).stdout
    size = {ln.split()[-1]: int(ln.split()[1], 16) for ln in out.splitlines() if len(ln.split()) == 4}
    n2, n3 = 65 * 49, 65 * 49 * 21
    assert size["blk3d_"] == 40 * n3 * 8 and size["blk2d_"] == 73 * n2 * 8
    assert size["blkcon_"] == 376 and size["blk1d_"] == 4 * 21 * 8 and size["blksiz_"] == 32
    # every hot-path routine of the reference is defined under its own (mangled) name
    out = subprocess.run(["nm", os.path.join(FDIR, "pom_gpu_host.o")], capture_output=True, text=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    for name in ("advave advct advq advt1 advt2 advu advv baropg dens profq proft profu profv vertvl realvertvl "
                 "bcond bcondorl lateral_viscosity mode_interaction mode_external mode_internal check_velocity").split():
        assert name + "_" in defined, name


@needs_flang
@pytest.mark.gpu
def test_fortran_driver_matches_oracle(tmp_path):
    from oracle.pyoracle import OracleTile, oracle_finish_initial
    _build()
    nsteps = 5
    a = make_case("island", 65, 49, 21, dte=6.0, isplit=30)
    oracle_finish_initial(a)
    with open(tmp_path / "state.in", "wb") as f:
        np.array([a.im, a.jm, -1, -1, -1, -1, nsteps, len(a.restore_records), a.bdry.size], dtype="<i4").tofile(f)
        for blk in (a.blk1d, a.blk2d, a.blk3d, a.bdry):
            blk.tofile(f)
        f.write(a.con.tobytes())
        for tr, sr in a.restore_records:
            np.ascontiguousarray(tr).tofile(f)
            np.ascontiguousarray(sr).tofile(f)
    (tmp_path / "pom.nml").write_text("&pom_nml\n title = 'island'\n netcdf_file = 'nonetcdf'\n mode = 3\n nadv = 2\n"
                                      " nitera = 1\n sw = 0.5\n npg = 1\n dte = 6.\n isplit = 30\n days = 1\n/\n")
    r = subprocess.run([os.path.join(FDIR, "pom_gpu_main"), "state.in", "state.out"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "error_status   0" in r.stdout
    OracleTile(a).run(nsteps)
    raw = np.fromfile(tmp_path / "state.out", dtype="<f8")
    n2, n3 = a.blk2d.size, a.blk3d.size
    b2 = raw[:n2].reshape(a.blk2d.shape)
    b3 = raw[n2:n2 + n3].reshape(a.blk3d.shape)
    bad = [n for i, n in enumerate(BLK2D) if n not in SCRATCH and not np.array_equal(a.blk2d[i], b2[i])]
    bad += [n for i, n in enumerate(BLK3D) if n not in SCRATCH and not np.array_equal(a.blk3d[i], b3[i])]
    assert not bad, bad
    # print_section's sums came from the device through the Fortran domain_stats wrapper (no state download)
    import ctypes
    line = [l for l in r.stdout.splitlines() if l.startswith("domain_stats:")][0]
    got = np.array([float(x) for x in line.split()[1:]])
    out = (ctypes.c_double * 8)()
    OracleTile(a).call("domain_stats", out, ctypes.c_int(0))
    np.testing.assert_allclose(got, np.array(list(out)), rtol=1e-12, atol=0)


# ---- the Fortran host on several MPI ranks, with file forcing and the output writers (pom_gpu_mpi_main) ----------------------
MPIEXEC = "/opt/conda/bin/mpiexec"
needs_mpi = pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists("/opt/conda/include/mpif.h")), reason="no MPI (MPICH) in this image")


def _build_mpi(iml, jml, nproc):
    import __graft_entry__ as ge
    ge.build_hip()
    subprocess.check_call(["make", "-C", FDIR, "mpi", "IM=65", "JM=49", "KB=21", f"IML={iml}", f"JML={jml}", f"NP={nproc}"], stdout=subprocess.DEVNULL)
    return os.path.join(FDIR, f"mpi_65x49x21_{iml}x{jml}p{nproc}", "pom_gpu_mpi_main")


def _write_state(path, st, nsteps, checks, wfiles=0, forced=False):
    """what pom_gpu_mpi_main reads for one rank: header, the COMMON blocks, the restore / forcing / lateral records"""
    nfrc = len(st.forcing_records["wind"]) if forced else 0
    nlat = len(st.lateral_records) if forced else 0
    hdr = [st.im, st.jm, st.n_west, st.n_east, st.n_south, st.n_north, nsteps, len(st.restore_records), st.bdry.size,
           st.i_off + 1, st.j_off + 1, nfrc, nlat, len(checks), wfiles, 0]
    with open(path, "wb") as f:
        np.array(hdr, dtype="<i4").tofile(f)
        np.array((list(checks) + [0] * 8)[:8], dtype="<i4").tofile(f)
        for blk in (st.blk1d, st.blk2d, st.blk3d, st.bdry):
            blk.tofile(f)
        f.write(st.con.tobytes())
        for tr, sr in st.restore_records:
            np.ascontiguousarray(tr).tofile(f)
            np.ascontiguousarray(sr).tofile(f)
        if forced:
            for kind in ("wind", "heat", "surface"):
                for a, b in st.forcing_records[kind]:
                    np.ascontiguousarray(a, dtype="<f8").tofile(f)
                    np.ascontiguousarray(b, dtype="<f8").tofile(f)
            for rec in st.lateral_records:
                for a in rec:
                    np.ascontiguousarray(a, dtype="<f8").tofile(f)


def _read_dump(path, st):
    """the blocks pom_gpu_mpi_main wrote after a listed step, into `st`"""
    raw = np.fromfile(path, dtype="<f8")
    n2, n3, nb = st.blk2d.size, st.blk3d.size, st.bdry.size
    st.blk2d[...] = raw[:n2].reshape(st.blk2d.shape)
    st.blk3d[...] = raw[n2:n2 + n3].reshape(st.blk3d.shape)
    st.bdry[...] = raw[n2 + n3:n2 + n3 + nb]
    return st


def _digest(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<f8").tobytes()).hexdigest()


NML = ("&pom_nml\n title = 'fortran host'\n netcdf_file = 'pomout'\n write_rst_file = 'pomrst'\n wrk_pth = '{wrk}/'\n time_start = '2000-01-01 00:00:00 +00:00'\n"
       " mode = 3\n nadv = 2\n nitera = 1\n sw = 0.5\n npg = 1\n dte = {dte}\n isplit = {isplit}\n days = {days}\n prtd1 = {prtd1}\n/\n")


@needs_flang
@needs_mpi
def test_fortran_mpi_driver_and_mover_build():
    """libpomgpu_mpi.so exports the mover's entry point; the multi-rank driver links every wrapper file an integrator links
    (hot path, ranks, forcing, I/O) and leaves no reference routine of the hot path undefined"""
    exe = _build_mpi(34, 26, 4)
    out = subprocess.run(["nm", "-D", os.path.join(ROOT, "extpom_amd", "csrc", "libpomgpu_mpi.so")], capture_output=True, text=True).stdout
    assert " T pomgpu_mpi_mover_install" in out and " T pomgpu_mpi_mover_remove" in out
    out = subprocess.run(["nm", exe], capture_output=True, text=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    for name in ("wind heat surface lateral_bc write_output_pnetcdf write_restart_pnetcdf sum0d_mpi bcast0d_mpi pomgpu_host_connect_mpi "
                 "pomgpu_host_neighbours pomgpu_host_finalize mode_internal advct profq").split():
        assert name + "_" in defined, name


@needs_flang
@needs_mpi
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["seamount_2x2_isplit10", "island_2x2"])
def test_fortran_mpi_ranks_hash_to_the_references_own_mpi_run(tmp_path, name):
    """Four MPI ranks of the FORTRAN host on one GPU (pom_gpu_mpi_main: pomgpu_host_connect_mpi -> the neighbour arithmetic of
    pomgpu_host_neighbours -> libpomgpu_mpi.so's MPI mover -> pomgpu_set_wide_external), every rank's restart-list fields over
    its (jm, im) cells -- ghost cells included -- against the digests of the reference's own four-process MPICH run
    (tests/golden/tiles_65x49x21_2x2.json; parallel_mpi.f:34-351).  isplit = 10: the wide-halo external mode with eight of a step's
    ten rounds on the second stream; isplit = 30 (island): the tiles are narrower than w + 3, pomgpu_set_wide_external declines
    on every rank and the per-point exchanges serve the external mode."""
    import json
    from extpom_amd import decomp
    from extpom_amd.cases import finish_initial
    from oracle.pyoracle import OracleTile
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "tiles_65x49x21_2x2.json")))
    (IM, JM, KB), (IML, JML), world = gold["grid"], gold["local"], gold["n_proc"]
    cfg = gold["configs"][name]
    exe = _build_mpi(IML, JML, world)
    checks = sorted(int(s) for s in cfg["steps"])
    tiles, states = [], []
    for r in range(world):
        tile = decomp.make_tile(r, IM, JM, IML, JML, n_proc=world)
        st = make_case(cfg["case"], IM, JM, KB, tile=tile, **cfg["nml"])
        ot = OracleTile(st)                                   # dens / baropg of the initialisation (npg = 1: no exchange inside)
        finish_initial(st, lambda s, a, b, c: ot.call("dens", ot.a3(a), ot.a3(b), ot.a3(c)), lambda s: ot.call("baropg"))
        _write_state(tmp_path / f"state.in.{r}", st, checks[-1], checks)
        tiles.append(tile); states.append(st)
    (tmp_path / "pom.nml").write_text(NML.format(wrk=tmp_path, dte=cfg["nml"]["dte"], isplit=cfg["nml"]["isplit"], days=1, prtd1=1))
    r = subprocess.run([MPIEXEC, "-n", str(world), exe, "state.in", "state.out"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "error_status   0" in r.stdout, r.stdout[-2000:]
    bad = []
    for step in checks:
        for rk in range(world):
            st = _read_dump(tmp_path / f"state.out.{rk}.{step}", states[rk])
            want = cfg["steps"][str(step)][rk]
            bad += [(step, rk, f) for f in gold["fields"] if _digest(st.field(f)[..., :tiles[rk].jm, :tiles[rk].im]) != want[f]]
    if bad:                                                   # where and by how much: against the single-tile oracle (owned and ghost cells)
        from oracle.pyoracle import oracle_finish_initial
        g = make_case(cfg["case"], IM, JM, KB, **cfg["nml"])
        oracle_finish_initial(g)
        OracleTile(g).run(bad[0][0])
        where = []
        for step, rk, f in [b for b in bad if b[0] == bad[0][0]][:16]:
            t = tiles[rk]
            got = _read_dump(tmp_path / f"state.out.{rk}.{step}", states[rk]).field(f)[..., :t.jm, :t.im]
            ref = g.field(f)[..., t.j_off:t.j_off + t.jm, t.i_off:t.i_off + t.im]
            d = np.argwhere(got != ref)
            where.append((step, rk, f, len(d), d[:4].tolist(), float(np.abs(got - ref).max())))
        raise AssertionError(f"{len(bad)} digests differ; against the single-tile oracle: {where}\n{r.stdout[-1500:]}")
    line = [l for l in r.stdout.splitlines() if l.startswith("message rounds per step")][0].split()
    total, side = int(line[line.index("total") + 1]), int(line[-1])
    if cfg["nml"]["isplit"] == 10:
        # wide mode under the reference's own call sequence: per step 2 rounds between kernels (advx + advy + aam, the late part of the wide
        # exchange) and 8 beside them on the second stream (the early part of the wide exchange, advct's edge lines, wr, and -- not in the
        # first step, which skips mode_internal's 3-D body -- w, the turbulence arrays, T / S / rho and the two velocity rounds that end
        # mode_internal: pomgpu_api.hip, "rim rounds") (+ the one-off static gather)
        assert side == 8 * checks[-1] - 5 and total < 4 * checks[-1] + 30, (total, side)
    else:                                                     # tiles too narrow for w = 34: ~200 rounds per step, none on the second stream
        assert side == 0 and total > 150 * checks[-1], (total, side)


@needs_flang
@needs_mpi
@pytest.mark.gpu
def test_fortran_forcing_and_io_wrappers_run_once(tmp_path):
    """pom_gpu_forcing.f90 (wind, heat, surface, lateral_bc: the reference's own decisions about WHEN a record is read,
    bounds_forcing.f:593-983, its readers served from records that came with the state) and pom_gpu_io.f90
    (write_output_pnetcdf / write_restart_pnetcdf: the reference's file names, advance.f:35-49) linked into one executable and
    RUN: the state hashes to the digests of the reference's OWN `advance` with file forcing (tests/golden/
    forced_advance_65x49x21.json) across record changes; the two files are read back with scipy and hold the final state."""
    import json
    from scipy.io import netcdf_file
    from extpom_amd.cases import make_forcing_records, make_lateral_records
    from extpom_amd.layout import RESTART_2D, RESTART_3D
    from oracle.pyoracle import oracle_finish_initial
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "forced_advance_65x49x21.json")))
    cfg = gold["config"]
    im, jm, kb = gold["grid"]
    exe = _build_mpi(im, jm, 1)
    st = make_case(cfg["case"], im, jm, kb, **cfg["nml"])
    oracle_finish_initial(st)
    make_forcing_records(st, cfg["forcing_records"])
    make_lateral_records(st, cfg["lateral_records"])
    checks = sorted(int(s) for s in cfg["steps"])
    _write_state(tmp_path / "state.in.0", st, checks[-1], checks, wfiles=1, forced=True)
    os.makedirs(tmp_path / "out")
    nml = cfg["nml"]
    (tmp_path / "pom.nml").write_text(NML.format(wrk=tmp_path, dte=nml["dte"], isplit=nml["isplit"], days=nml["days"], prtd1=nml["prtd1"]))
    r = subprocess.run([MPIEXEC, "-n", "1", exe, "state.in", "state.out"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "error_status   0" in r.stdout, r.stdout[-2000:]
    bad = []
    for step in checks:
        _read_dump(tmp_path / f"state.out.0.{step}", st)
        want = cfg["steps"][str(step)]
        bad += [(step, f) for f in gold["fields"] if _digest(st.field(f)) != want[f]]
        if _digest(st.bdry) != want["bdry"]:
            bad.append((step, "bdry"))
    assert not bad, bad[:12]
    # the files: names as the reference builds them (nprint = iint / iprint = 0 here, advance.f:38,46), contents = the final state
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cdf_check import check_header
    check_header(tmp_path / "out" / "pomout.0000.nc", "output", "fortran host", "2000-01-01 00:00:00 +00:00", kb, im, jm)
    check_header(tmp_path / "out" / "pomrst.0000.nc", "restart", "fortran host", "2000-01-01 00:00:00 +00:00", kb, im, jm)
    with netcdf_file(str(tmp_path / "out" / "pomout.0000.nc"), "r", mmap=False) as f:
        assert f.title == b"fortran host" and f.variables["time"].units == b"days since 2000-01-01 00:00:00 +00:00"
        for n in ("uab", "vab", "elb"):
            assert np.array_equal(f.variables[n][0], st.field(n)), n
        for n in ("u", "v", "t", "s", "rho"):
            assert np.array_equal(f.variables[n][0], st.field(n)[:kb - 1]), n
    with netcdf_file(str(tmp_path / "out" / "pomrst.0000.nc"), "r", mmap=False) as f:
        assert float(f.variables["iint"].getValue()) == float(checks[-1])
        for n in RESTART_2D + RESTART_3D:
            assert np.array_equal(f.variables[n][:], st.field(n)), n
