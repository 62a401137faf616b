"""Deterministic idealised cases: the inputs the reference reads from netCDF files.

The reference ships no input data and no case generator (its ``read_grid`` /
``initial_conditions`` read PnetCDF files, reference pom/initialize.f:317-463), so parity is judged
on inputs generated here and fed identically to the reference build, the CPU oracle and the HIP
path.  ``make_case`` fills what the readers would fill *plus* the quantities ``read_grid`` derives on
the host (initialize.f:329-389; u/v masks as io_pnetcdf.F:2243-2256).  ``finish_initial`` then runs
the rest of the reference's initialisation sequence (``initial_conditions`` after the reads,
``update_initial``, ``bottom_friction``: initialize.f:413-544) with the equation of state and the
baroclinic pressure gradient supplied by the caller -- the reference calls its own ``dens`` /
``baropg`` there, and each backend does likewise.

Cases
  seamount : Gaussian seamount in a 4500 m channel, closed N/S walls, open E/W with 0.2 m/s
             through-flow, T = 5 + 15 exp(z/1000), S = 35 (the classic POM test problem).
  basin    : closed basin (land rim), gently sloping bottom with bumps, linear stratification plus
             a seeded (12345) 1e-3 degC perturbation, steady zonal wind stress.
  island   : open on all four sides around a shoaling bank (exercises N/S open-boundary code).
"""
from __future__ import annotations

import math

import numpy as np

from .layout import PomState
from .namelist import apply_constants, run_constants


def sigma_levels(kb: int) -> tuple[np.ndarray, np.ndarray]:
    """Sigma grid: geometric refinement towards surface and bottom, uniform in between."""
    nlog = max(1, min(4, (kb - 3) // 4))
    w = np.ones(kb - 1)
    for n in range(nlog):
        w[n] = 0.5 ** (nlog - n)
        w[kb - 2 - n] = 0.5 ** (nlog - n)
    w /= w.sum()
    z = np.zeros(kb)
    z[1:] = -np.cumsum(w)
    z[kb - 1] = -1.0
    zz = np.zeros(kb)
    zz[:kb - 1] = 0.5 * (z[:kb - 1] + z[1:])
    zz[kb - 1] = 2.0 * zz[kb - 2] - zz[kb - 3]
    return z, zz


def _hash_noise(gi, gj, k, seed=12345):
    """Deterministic, decomposition-independent pseudo-noise in [-1,1) from GLOBAL indices."""
    x = np.sin(gi * 12.9898 + gj * 78.233 + k * 37.719 + seed * 0.0001) * 43758.5453
    return 2.0 * (x - np.floor(x)) - 1.0


def make_case(name: str, im: int, jm: int, kb: int, tile=None, **nml) -> PomState:
    """State of one tile (default: the whole im x jm grid as a single tile) holding everything the
    reference's readers + read_grid provide.  Every value is a function of GLOBAL indices and 1-D
    global coordinate vectors, so a tile generated here equals the same window cut out of the
    single-tile state bit for bit (tests/test_host_logic.py) without any rank ever building the
    global 3-D arrays."""
    if tile is None:
        from .decomp import make_tile
        tile = make_tile(0, im, jm, im, jm)
    st = PomState(tile.im_local, tile.jm_local, kb, tile.im, tile.jm)
    st.n_west, st.n_east, st.n_south, st.n_north = tile.n_west, tile.n_east, tile.n_south, tile.n_north
    st.i_off, st.j_off = tile.i_off, tile.j_off
    c = run_constants(None, **nml)
    apply_constants(st, c)
    z, zz = sigma_levels(kb)
    st.z[...] = z
    st.zz[...] = zz
    st.dz[:kb - 1] = z[:kb - 1] - z[1:]
    st.dzz[:kb - 1] = zz[:kb - 1] - zz[1:]

    # ---- global 1-D coordinate vectors (index 0 = global point 1) -------------------------
    delx = 8000.0
    stretch = 0.5 if name == "basin" else 1.0
    gi1 = np.arange(1, im + 1, dtype=np.float64)
    gj1 = np.arange(1, jm + 1, dtype=np.float64)
    dx1 = delx - stretch * delx * np.sin(math.pi * gi1 / im) / 2.0
    dy1 = delx - stretch * delx * np.sin(math.pi * gj1 / jm) / 2.0
    ec1 = np.concatenate(([0.0], np.cumsum(dx1[:-1])))
    nc1 = np.concatenate(([0.0], np.cumsum(dy1[:-1])))
    ee1 = ec1 + 0.5 * dx1
    ne1 = nc1 + 0.5 * dy1
    lx, ly = float(ee1[-1]), float(ne1[-1])
    x0, y0 = float(ee1[(im + 1) // 2 - 1]), float(ne1[(jm + 1) // 2 - 1])

    # ---- window = tile plus one extra column/row on the low side (for the u/v masks) ------
    ti, tj, io, jo = tile.im, tile.jm, tile.i_off, tile.j_off
    wi = np.arange(io, io + ti + 1)          # global i (1-based) of window columns; wi[0] may be 0
    wj = np.arange(jo, jo + tj + 1)
    ci = np.clip(wi, 1, im) - 1              # index into the 1-D vectors
    cj = np.clip(wj, 1, jm) - 1
    X = ee1[ci][None, :]
    Y = ne1[cj][:, None]
    xc, yc = X - x0, Y - y0
    vel = 0.0
    if name == "seamount":
        ra = 0.12 * min(lx, ly) + 12500.0
        hw = 4500.0 * (1.0 - 0.9 * np.exp(-(xc * xc + yc * yc) / (ra * ra)))
        vel = 0.2
    elif name == "island":
        ra = 0.10 * min(lx, ly) + 10000.0
        hw = 3000.0 * (1.0 - 0.85 * np.exp(-(xc * xc + 1.7 * yc * yc) / (ra * ra)))
        vel = 0.1
    elif name == "basin":
        sx, sy = X / lx, Y / ly
        hw = (1500.0 + 1500.0 * sx + 400.0 * np.sin(3.0 * math.pi * sx) * np.sin(2.0 * math.pi * sy)
              + 250.0 * np.cos(7.0 * math.pi * sy))
    else:
        raise ValueError(f"unknown case {name!r}")
    hw = np.where(hw < 1.0, 1.0, hw) + 0.0 * X
    gI = wi[None, :] + 0 * wj[:, None]
    gJ = wj[:, None] + 0 * wi[None, :]
    if name in ("seamount", "basin"):
        hw = np.where((gJ <= 1) | (gJ >= jm), 1.0, hw)          # closed north/south walls
    if name == "basin":
        hw = np.where((gI <= 1) | (gI >= im), 1.0, hw)          # land rim all round
    fw = np.where(hw > 1.0, 1.0, 0.0)
    # masks (io_pnetcdf.F:2243-2256): a u/v point is closed when the cell behind it is land
    dumw = fw.copy()
    dvmw = fw.copy()
    dumw[:, 1:][(fw[:, :-1] == 0.0) & (fw[:, 1:] != 0.0)] = 0.0
    dvmw[1:, :][(fw[:-1, :] == 0.0) & (fw[1:, :] != 0.0)] = 0.0
    if io == 0:      # global column 1 has nothing behind it
        dumw[:, 1] = fw[:, 1]
    if jo == 0:
        dvmw[1, :] = fw[1, :]

    A = (slice(0, tj), slice(0, ti))         # active part of the (padded) tile arrays
    W = (slice(1, None), slice(1, None))     # the tile inside the window
    st.h[A] = hw[W]
    st.fsm[A] = fw[W]
    st.dum[A] = dumw[W]
    st.dvm[A] = dvmw[W]
    li, lj = ci[1:], cj[1:]
    st.dx[A] = dx1[li][None, :] + 0.0 * dy1[lj][:, None]
    st.dy[A] = dy1[lj][:, None] + 0.0 * dx1[li][None, :]
    st.east_c[A] = ec1[li][None, :] + 0.0 * nc1[lj][:, None]
    st.north_c[A] = nc1[lj][:, None] + 0.0 * ec1[li][None, :]
    st.east_e[A] = st.east_c[A] + 0.5 * st.dx[A]
    st.north_e[A] = st.north_c[A] + 0.5 * st.dy[A]
    st.east_u[A] = st.east_c[A]
    st.north_u[A] = st.north_c[A] + 0.5 * st.dy[A]
    st.east_v[A] = st.east_c[A] + 0.5 * st.dx[A]
    st.north_v[A] = st.north_c[A]
    # read_grid's derived quantities (initialize.f:329-386)
    lat = 30.0 + 15.0 * st.north_e[A] / ly
    st.cor[A] = 2.0 * 7.29e-5 * np.sin(lat * (st.pi / 180.0))
    lat_mid = 30.0 + 15.0 * ne1[jm // 2 - 1] / ly
    st.period = (2.0 * st.pi) / abs(2.0 * 7.29e-5 * math.sin(lat_mid * (st.pi / 180.0))) / 86400.0
    st.art[A] = st.dx[A] * st.dy[A]
    dxw, dyw = dx1[ci], dy1[cj]              # window-indexed spacings
    # aru/arv for global i,j >= 2; row/column 1 copy their neighbour (initialize.f:362-378)
    aruw = 0.25 * (dxw[None, 1:] + dxw[None, :-1]) * (dyw[1:, None] + dyw[1:, None]) + 0.0
    arvw = 0.25 * (dxw[None, 1:] + dxw[None, 1:]) * (dyw[1:, None] + dyw[:-1, None]) + 0.0
    st.aru[A] = aruw
    st.arv[A] = arvw
    if io == 0:
        st.aru[:tj, 0] = st.aru[:tj, 1]
        st.arv[:tj, 0] = st.arv[:tj, 1]
    if jo == 0:
        st.aru[0, :ti] = st.aru[1, :ti]
        st.arv[0, :ti] = st.arv[1, :ti]
    st.d[...] = st.h + st.el
    st.dt[...] = st.h + st.et

    # ---- initial T/S and climatology (level kb defined explicitly -- the reference's reader
    # leaves it uninitialised, io_pnetcdf.F:2825-2839) ------------------------------------
    hA = st.h[A]
    depth = zz[:, None, None] * hA[None]
    A3 = (slice(None),) + A
    if name == "basin":
        kk = np.arange(1, kb + 1, dtype=np.float64)[:, None, None]
        noise = _hash_noise(gI[W][None].astype(np.float64), gJ[W][None].astype(np.float64), kk)
        st.tclim[A3] = 18.0 + 14.0 * depth / 3500.0
        st.tb[A3] = st.tclim[A3] + 1.0e-3 * noise
        st.sb[A3] = 35.0 - 0.6 * depth / 3500.0
        st.sclim[A3] = st.sb[A3]
    else:
        st.tb[A3] = 5.0 + 15.0 * np.exp(depth / 1000.0) - st.tbias
        st.sb[A3] = 35.0 - st.sbias
        st.tclim[A3] = st.tb[A3]
        st.sclim[A3] = st.sb[A3]
    for f in (st.tb, st.sb, st.tclim, st.sclim):
        f[kb - 1] = f[kb - 2]
        f[...] *= st.fsm[None]

    # ---- initial flow and open-boundary data ------------------------------------------------
    st.ub[:kb - 1] = vel * st.dum[None]
    st.uab[...] = vel * st.dum
    st.uabe[...] = st.uab[:, ti - 2]
    st.uabw[...] = st.uab[:, 1]
    if name == "island":
        st.vb[:kb - 1] = 0.25 * vel * st.dvm[None]
        st.vab[...] = 0.25 * vel * st.dvm
        st.vabn[...] = st.vab[tj - 2, :]
        st.vabs[...] = st.vab[1, :]
        st.vabe[...] = st.vab[:, ti - 1]
        st.vabw[...] = st.vab[:, 0]
        st.uabn[...] = st.uab[tj - 1, :]
        st.uabs[...] = st.uab[0, :]
        st.vbn[...] = st.vb[:, tj - 2, :]
        st.vbs[...] = st.vb[:, 1, :]
        st.ubn[...] = st.ub[:, tj - 1, :]
        st.ubs[...] = st.ub[:, 0, :]
    st.ube[...] = st.ub[:, :, ti - 2]
    st.ubw[...] = st.ub[:, :, 1]

    # ---- surface forcing, constant in time (the reference's wind/heat readers are out of scope)
    if name == "basin":
        st.wusurf[A] = -1.0e-4 * np.cos(math.pi * st.north_e[A] / ly) * st.fsm[A]
        st.wtsurf[A] = 2.0e-6 * np.sin(2.0 * math.pi * st.east_e[A] / lx) * st.fsm[A]
    else:
        st.wusurf[A] = -2.0e-5 * st.fsm[A]
        st.wvsurf[A] = 1.0e-5 * np.sin(math.pi * st.east_e[A] / lx) * st.fsm[A]

    # relaxation targets served to restore_interior (bounds_forcing.f:1039-1065 reads record
    # iint/irst+1 at iint=2 and the following one): climatology, the second record 0.01 degC /
    # 0.002 psu warmer/saltier so that the time interpolation is exercised.
    st.restore_records = [
        (st.tclim[A3].copy(), st.sclim[A3].copy()),
        ((st.tclim[A3] + 0.01) * st.fsm[A][None], (st.sclim[A3] + 0.002) * st.fsm[A][None]),
    ]
    return st


def finish_initial(st: PomState, dens, baropg) -> PomState:
    """Remainder of the reference's initialisation (initialize.f:413-544) on a filled state.

    ``dens(st, si, ti, rho)`` and ``baropg(st)`` are the caller's implementations of the reference
    routines of the same name (solver.f:1162-1209, :848-940); every array argument is a field NAME.
    """
    kb, im, jm = st.kb, st.im, st.jm
    dens(st, "sclim", "tclim", "rmean")                    # initialize.f:413
    dens(st, "sb", "tb", "rho")                            # initialize.f:422
    st.tsurf[...] = st.tb[0]
    st.ssurf[...] = st.sb[0]
    st.rfe = st.rfw = st.rfn = st.rfs = 1.0
    st.tbe[:kb - 1] = st.tb[:kb - 1, :, im - 1]
    st.tbw[:kb - 1] = st.tb[:kb - 1, :, 0]
    st.sbe[:kb - 1] = st.sb[:kb - 1, :, im - 1]
    st.sbw[:kb - 1] = st.sb[:kb - 1, :, 0]
    st.tbn[:kb - 1] = st.tb[:kb - 1, jm - 1, :]
    st.tbs[:kb - 1] = st.tb[:kb - 1, 0, :]
    st.sbn[:kb - 1] = st.sb[:kb - 1, jm - 1, :]
    st.sbs[:kb - 1] = st.sb[:kb - 1, 0, :]
    # update_initial (initialize.f:466-521)
    st.ua[...] = st.uab
    st.va[...] = st.vab
    st.el[...] = st.elb
    st.et[...] = st.etb
    st.etf[...] = st.et
    st.d[...] = st.h + st.el
    st.dt[...] = st.h + st.et
    st.w[0] = st.vfluxf
    st.l[...] = float(np.float32(0.1)) * st.dt[None]       # "0.1*dt": REAL(4) literal (initialize.f:484)
    st.q2b[...] = st.small
    st.q2lb[...] = st.l * st.q2b
    st.kh[...] = st.l * np.sqrt(st.q2b)
    st.km[...] = st.kh
    st.kq[...] = st.kh
    st.aam[...] = st.aam_init
    st.q2[...] = st.q2b
    st.q2l[...] = st.q2lb
    st.t[...] = st.tb
    st.s[...] = st.sb
    st.u[...] = st.ub
    st.v[...] = st.vb
    baropg(st)
    for k in range(kb - 1):
        st.drx2d[...] += st.drhox[k] * st.dz[k]
        st.dry2d[...] += st.drhoy[k] * st.dz[k]
    # bottom_friction (initialize.f:524-544)
    with np.errstate(divide="ignore"):      # h = 0 only in the padding of a trimmed tile
        cbc = (st.kappa / np.log((1.0 + st.zz[kb - 2]) * st.h / st.z0b)) ** 2
    st.cbc[...] = np.minimum(st.cbcmax, np.maximum(st.cbcmin, cbc))
    return st


def cut_tile(g: PomState, tile) -> PomState:
    """Extract one tile (with its ghost rim) of a global single-tile state."""
    st = PomState(tile.im_local, tile.jm_local, g.kb, tile.im, tile.jm)
    i0, j0, im, jm = tile.i_off, tile.j_off, tile.im, tile.jm
    st.blk1d[...] = g.blk1d
    st.blk2d[:, :jm, :im] = g.blk2d[:, j0:j0 + jm, i0:i0 + im]
    st.blk3d[:, :, :jm, :im] = g.blk3d[:, :, j0:j0 + jm, i0:i0 + im]
    from .layout import BDRY
    for n, kind in BDRY:
        src, dst = g.field(n), st.field(n)
        if kind == "J":
            dst[:jm] = src[j0:j0 + jm]
        elif kind == "I":
            dst[:im] = src[i0:i0 + im]
        elif kind == "JK":
            dst[:, :jm] = src[:, j0:j0 + jm]
        else:
            dst[:, :im] = src[:, i0:i0 + im]
    st.con[...] = g.con
    st.n_west, st.n_east, st.n_south, st.n_north = tile.n_west, tile.n_east, tile.n_south, tile.n_north
    st.i_off, st.j_off = i0, j0
    if hasattr(g, "forcing_records"):
        st.forcing_records = {k: [(np.ascontiguousarray(a[j0:j0 + jm, i0:i0 + im]), np.ascontiguousarray(b[j0:j0 + jm, i0:i0 + im]))
                                  for a, b in v] for k, v in g.forcing_records.items()}
    if hasattr(g, "restore_records"):
        st.restore_records = [(np.ascontiguousarray(a[:, j0:j0 + jm, i0:i0 + im]),
                               np.ascontiguousarray(b[:, j0:j0 + jm, i0:i0 + im])) for a, b in g.restore_records]
    return st


def make_forcing_records(st: PomState, count: int = 4) -> PomState:
    """Synthetic records behind surface_forcing (bounds_forcing.f:871-983: wind and heat every 0.125 d,
    time-interpolated; surface = SST without interpolation): the constant fields of make_case modulated per
    record, so that the interpolation weights, the record shift and the masks all show up in the result.
    Opt-in: a state WITHOUT these records keeps its constant forcing (advance skips surface_forcing)."""
    A = (slice(0, st.jm), slice(0, st.im))
    ramp = lambda r, a: a * (1.0 + 0.25 * math.sin(0.9 * r)) * st.fsm[A]
    rr = range(1, count + 1)
    st.forcing_records = {
        "wind": [(ramp(r, st.wusurf[A]) - 1.0e-6 * r * st.fsm[A], ramp(r + 3, st.wvsurf[A])) for r in rr],
        "heat": [(ramp(r, st.wtsurf[A]) + 1.0e-7 * r * st.fsm[A], 1.0e-5 * (1.0 + 0.1 * r) * st.fsm[A]) for r in rr],
        "surface": [((st.t[0][A] + 0.05 * r) * st.fsm[A], st.s[0][A] * st.fsm[A]) for r in rr],
    }
    return st


LATERAL_ORDER = ("tbw", "sbw", "ubw", "vbw", "tbe", "sbe", "ube", "vbe", "tbn", "sbn", "vbn", "ubn", "tbs", "sbs", "vbs", "ubs",
                 "elw", "ele", "eln", "els")


def make_lateral_records(st: PomState, count: int = 4) -> PomState:
    """Synthetic records for lateral_bc (bounds_forcing.f:593-868): what read_boundary_conditions_pnetcdf would
    deliver for records 1..count -- the 20 arrays of LATERAL_ORDER in the shape of the bdry members they land in --
    derived from the boundary values of the finished initial state, modulated per record.  Call after
    finish_initial()."""
    recs = []
    for r in range(1, count + 1):
        rec = []
        for n in LATERAL_ORDER:
            a = st.field(n)
            if n.startswith("el"):
                x = a + 1.0e-3 * math.sin(0.7 * r) * (a.shape[0] > 0)
            elif n[0] in "ts":
                x = a * (1.0 + 0.002 * r)
            else:
                x = a * (1.0 + 0.2 * math.sin(1.3 * r)) + 1.0e-4 * r * (a != 0)
            rec.append(np.ascontiguousarray(x, dtype=np.float64))
        recs.append(rec)
    st.lateral_records = recs
    return st
