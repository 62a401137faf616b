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
