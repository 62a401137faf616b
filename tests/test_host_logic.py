"""Host-side logic that needs no GPU: namelist, decomposition arithmetic, case generator on tiles,
COMMON-block layout, C-ABI symbol table, 2-rank halo exchange over gloo."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from extpom_amd import decomp, namelist
from extpom_amd.cases import cut_tile, make_case
from extpom_amd.layout import BLK2D, BLK3D, CON_DTYPE, P2, P3, PomState

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_sizes_match_reference_common_blocks():
    # sizes verified with nm -S on the flang object (SURVEY 8a "types"): 73 2-D, 40 3-D arrays, blkcon 376 B
    assert len(BLK2D) == 73 and len(BLK3D) == 40 and CON_DTYPE.itemsize == 376
    assert (P3["aam"], P3["zflux"], P2["aam2d"], P2["wvsurff"]) == (0, 39, 0, 72)
    st = PomState(142, 306, 40)
    assert st.blk3d.nbytes == 556185600 and st.blk2d.nbytes == 73 * 142 * 306 * 8 and st.blk1d.nbytes == 1280
    assert st.u.shape == (40, 306, 142) and st.u.ctypes.data == st.blk3d.ctypes.data + P3["u"] * 8 * 142 * 306 * 40


def test_namelist_defaults_and_derived_constants(tmp_path):
    nml = tmp_path / "pom.nml"
    nml.write_text("&pom_nml\n  title = 'x' ! c\n  mode = 3\n  nadv = 2\n  dte = 2.\n  isplit = 30\n  days = 1\n  prtd1 = 0.1\n"
                   "  write_rst = 1.0\n  swtch = 9999.\n  netcdf_file = 'nonetcdf'\n/\n! trailing doc\n")
    c = namelist.read_namelist_file(str(nml))
    assert c["dti"] == 60.0 and c["dte2"] == 4.0 and c["dti2"] == 120.0
    assert c["iend"] == 1440 and c["iprint"] == 144 and c["irestart"] == 1440
    assert c["ispi"] == 1.0 / 30.0 and c["isp2i"] == 1.0 / 60.0
    assert (c["horcon"], c["tprni"], c["smoth"], c["nbct"], c["ispadv"]) == (0.1, 0.1, 0.1, 1, 1)
    with pytest.raises(ValueError):
        namelist.parse_namelist("&pom_nml\n bogus = 1\n/\n")


def test_decomposition_matches_distribute_mpi():
    # pom.h_dist's own example: 282x306 in tiles of 142x306 -> 2x1
    assert decomp.tile_grid(282, 306, 142, 306) == (2, 1)
    t0, t1 = decomp.make_tile(0, 282, 306, 142, 306), decomp.make_tile(1, 282, 306, 142, 306)
    assert (t0.n_west, t0.n_east, t1.n_west, t1.n_east) == (-1, 1, 0, -1)
    assert t1.i_off == 140 and t0.im == t1.im == 142      # neighbours share 2 columns
    # BASELINE configs: 1024x1024 on 2x4 -> 513x258 (north row trimmed to 256), 2048x1536 -> 1025x386 (384)
    assert decomp.local_size(1024, 1024, 2, 4) == (513, 258)
    assert decomp.make_tile(7, 1024, 1024, 513, 258).jm == 256
    assert decomp.local_size(2048, 1536, 2, 4) == (1025, 386)
    assert decomp.make_tile(6, 2048, 1536, 1025, 386).jm == 384
    with pytest.raises(ValueError):
        decomp.make_tile(0, 282, 306, 100, 306, n_proc=2)   # "im_local or jm_local is too low"
    for n in (1, 2, 4, 8):
        nx, ny = decomp.choose_tile_grid(n, 2048, 1536)
        assert (nx, ny) == (1, n)                           # whole rows while the tiles stay tall enough (profiles/round3_tile_grids.txt)
    assert decomp.choose_tile_grid(4, 65, 49) == (2, 2)


@pytest.mark.parametrize("case", ["seamount", "island", "basin"])
def test_tile_generation_equals_cut_of_global(case):
    g = make_case(case, 65, 49, 21, dte=6.0, isplit=30)
    for r in range(4):
        t = decomp.make_tile(r, 65, 49, 34, 26)
        a, b = make_case(case, 65, 49, 21, tile=t, dte=6.0, isplit=30), cut_tile(g, t)
        assert not [n for n in BLK2D + BLK3D if not np.array_equal(a.field(n), b.field(n))]
        for (x, y), (p, q) in zip(a.restore_records, b.restore_records):
            assert np.array_equal(x, p) and np.array_equal(y, q)


def test_c_abi_library_exports_every_declared_symbol():
    """libpomgpu.so loads without a GPU and exports exactly what include/pomgpu.h declares"""
    import re
    import __graft_entry__ as ge
    so = ge.build_hip()
    lib = ctypes.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "pomgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(pomgpu_[a-z0-9_]+)\s*\(", hdr)) - {"pomgpu_exchange_fn"})
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in pomgpu.h but not exported"
    from extpom_amd import lib as binding
    assert set(binding.EXPORTS) <= set(declared)
    # the fp32-storage study variant (BASELINE configs[4], same sources with -DPOMGPU_STORE_F32) exports the same C ABI
    lib32 = ctypes.CDLL(ge.build_hip(f32=True))
    for name in declared:
        assert hasattr(lib32, name), f"{name} missing from libpomgpu_f32.so"
    lib32.pomgpu_version.restype = lib.pomgpu_version.restype = ctypes.c_char_p
    assert b"fp32-storage" in lib32.pomgpu_version() and b"fp32" not in lib.pomgpu_version()
    # no device here: context creation must refuse, not fall back
    from extpom_amd.lib import Dims
    h = ctypes.c_void_p()
    d = Dims(65, 49, 21, 65, 49, -1, -1, -1, -1)
    lib.pomgpu_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(Dims), ctypes.c_int, ctypes.c_void_p]
    import torch
    if not torch.cuda.is_available():
        assert lib.pomgpu_create(ctypes.byref(h), ctypes.byref(d), 0, None) == -4   # POMGPU_ENODEV


def test_two_rank_halo_exchange_is_decomposition_invariant():
    """world_size-2 gloo run: the CPU oracle on a 2x1 (and 1x2) split, with extpom_amd.halo doing
    every exchange, must equal the single-tile run bit for bit"""
    for split in ("x", "y", "xy"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "halo_worker.py"), split],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "HALO-OK" in r.stdout, r.stdout + r.stderr


def test_order_exchange_of_baropg_mcc_is_decomposition_invariant():
    """npg = 2: the 4th-order pressure gradient needs one more ghost column / row (order2d_mpi,
    order3d_mpi); 2x2 split over gloo against the single-tile run"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "halo_worker.py"), "xy", "npg2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "HALO-OK" in r.stdout, r.stdout + r.stderr


def test_domain_stats_partial_sums_add_up_over_tiles():
    """sums_only: the tile-local sums of a 2x2 split (what sum0d_mpi would add) equal the single-tile sums"""
    import ctypes
    import numpy as np
    from extpom_amd import decomp
    from extpom_amd.cases import cut_tile, make_case
    from oracle.pyoracle import OracleTile, oracle_finish_initial
    g = make_case("island", 41, 35, 11, dte=6.0, isplit=10)
    oracle_finish_initial(g)
    OracleTile(g).run(3)

    def sums(st):
        out = (ctypes.c_double * 8)()
        OracleTile(st).call("domain_stats", out, ctypes.c_int(1))
        return np.array(list(out))

    whole = sums(g)
    iml, jml = decomp.local_size(41, 35, 2, 2)
    parts = sum(sums(cut_tile(g, decomp.make_tile(r, 41, 35, iml, jml, n_proc=4))) for r in range(4))
    np.testing.assert_allclose(parts, whole, rtol=1e-12, atol=0)


def test_host_libm_is_the_one_the_pow_clone_restates(tmp_path):
    """bit-parity with the reference leans on glibc's (not correctly rounded) pow: `dens` calls it for abs(sr)**1.5 and the
    HIP kernel restates glibc 2.35's algorithm (THIRD_PARTY_NOTICES.md).  A host with another libm changes what "the
    reference" computes -- so (i) the libc version is pinned here and stated on the bench line, (ii) the restatement
    (tools/check_glibc_pow_clone.c, the same code as gpow15 in k_adv.hip) equals this host's pow() on 2.2e7 arguments."""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.gnu_get_libc_version.restype = ctypes.c_char_p
    ver = libc.gnu_get_libc_version().decode()
    assert ver.startswith("2.35"), f"glibc {ver}: the pow tables were read out of 2.35 -- re-run tools/dump_glibc_pow_tables.py and the parity tests"
    exe = tmp_path / "chkpow"
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tools", "check_glibc_pow_clone.c"), "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300).stdout
    assert "x**1.5: 0 mismatches" in out and "general x**y: 0 mismatches" in out, out


def _bench_selftest(plan, gpus, extra=(), limit="10"):
    """bench.py --gpus N with stand-in ranks (POM_BENCH_SELFTEST: no GPU; what the rank SUPERVISORS make of passes that
    complete, hang or die is what runs here -- the real code path of the launcher, the supervisors and their rendezvous)"""
    import json
    env = dict(os.environ, POM_BENCH_REHEARSE="1", POM_BENCH_SELFTEST=json.dumps(plan), POM_BENCH_PASS_LIMIT=limit)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


def test_bench_multi_gpu_passes_survive_a_hang_and_a_dead_rank():
    """north_star's multi-GPU line must not be losable: the ranks' supervisors run a pass with every round on one stream, a
    pass with the second stream and communicator, and a pass on the alternate tile grid, each as fresh child processes.  A
    pass whose rank hangs (here: rank 1 of the overlapped pass never posts) is ended at its limit and named, a pass whose rank
    dies (rank 2 of the alternate-grid pass) ends at once on every rank -- and the line still carries the passes that completed,
    ONE line, with the tile grids BASELINE names (2x4 beside 1x8 would be --gpus 8: four ranks here)."""
    # (limit: a healthy stand-in pass is four python processes importing torch and meeting on gloo -- 2-3 s on an idle box, more on a busy one)
    r, lines = _bench_selftest({"hang": [1, 1], "fail": [2, 2]}, 4, ["--workload", "basin2048"], limit="15")
    assert r.returncode == 0 and len(lines) == 1, r.stdout + r.stderr
    out = lines[0]
    ps = out["passes"]
    assert [(p["tiles"], p["overlap"], p["ok"]) for p in ps] == [("1x4", False, True), ("1x4", True, False), ("2x2", False, False)], ps
    assert out["config"]["tiles"] == "1x4" and out["config"]["tiles_primary"] == "1x4" and ps[0]["primary"]
    assert out["config"]["no_overlap_env"] == "1"                       # pass 0 ran with POMGPU_NO_OVERLAP=1
    assert "no result within" in ps[1]["failed"]["first"] and set(ps[1]["failed"]["phase_by_rank"].values()) == {"connect"}
    assert ps[2]["failed"]["why_by_rank"]["2"] == "exit code 9" and ps[2]["wall_s"] < 12.0     # the dead rank ended the pass at once, not at the limit (15 s)
    # everything healthy: the faster pass (overlap on, in the stand-in's numbers) is the line's value, the alternate grid under it
    r, lines = _bench_selftest({}, 2, ["--workload", "basin1024", "--tiles", "2x1"])
    assert r.returncode == 0 and len(lines) == 1, r.stdout + r.stderr
    out = lines[0]
    assert [(p["tiles"], p["overlap"], p["ok"]) for p in out["passes"]] == [("2x1", False, True), ("2x1", True, True), ("1x2", True, True)]
    assert out["config"]["overlap"] is True and out["ms_per_step"] == 8.0 and out["passes"][1]["primary"]
    # nothing completes: no line, non-zero exit
    r, lines = _bench_selftest({"fail": [0, 0]}, 2, ["--only-pass", "0"])
    assert r.returncode != 0 and not lines


def test_bench_tile_grids_follow_baseline_configs():
    """configs[2] names "2x4 tile decomposition on 8 GPUs" for 1024x1024x40; configs[3] (2048x1536x50) names none: whole rows
    there.  The other grid is always measured beside the primary (parallel_mpi.f:54-65 leaves the split to im_local, jm_local)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.tile_grids("basin1024", 8) == ("2x4", "1x8")
    assert bench.tile_grids("basin2048", 8) == ("1x8", "2x4")
    assert bench.tile_grids("basin2048", 8, "2x4") == ("2x4", "1x8")
    assert bench.tile_grids("basin2048", 4) == ("1x4", "2x2") and bench.tile_grids("basin2048", 2)[0] == "1x2"
    with pytest.raises(SystemExit):
        bench.tile_grids("basin2048", 8, "3x3")


def test_short_wave_term_in_double_double_equals_the_quad_evaluation(tmp_path):
    """csrc/dd_exp.h (what the proft kernels evaluate for solver.f:1608-1611) against the REAL(16) expression as libquadmath
    evaluates it, on arguments of proft's ranges (tools/check_dd_exp.cpp; 2e7 arguments there, 3e5 here): no result may differ."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "check_dd_exp")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_dd_exp.cpp"), "-lquadmath"])
    r = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 results differ" in r.stdout, r.stdout[-500:]


@pytest.mark.bg_oracle("seamount", 33, 29, 9, 2)
def test_background_oracle_leaves_the_digests_of_an_in_process_run(bg_oracle):
    """tests/oracle_bg.py + tests/conftest.py: the child process that computes the full-size oracle steps beside the GPU tests
    (test_config4_2048x1536x50_full_size joins it) -- on a small grid its per-array digests after every step are those of the
    same oracle run made here, and a changed bit (the sign of a zero) changes a digest"""
    from oracle_bg import NML, digests
    from extpom_amd.cases import make_case
    from oracle.pyoracle import OracleTile, oracle_finish_initial
    a = make_case("seamount", 33, 29, 9, **NML)
    oracle_finish_initial(a)
    assert bg_oracle.step(0)["digests"] == digests(a)
    oc = OracleTile(a)
    for n in (1, 2):
        oc.run(1)
        want = bg_oracle.step(n, timeout=120)
        assert want["iint"] == n and want["digests"] == digests(a)
    a.field("el")[0, 0] = -0.0 if a.field("el")[0, 0] == 0 else -a.field("el")[0, 0]
    assert digests(a)["el"] != want["digests"]["el"]
