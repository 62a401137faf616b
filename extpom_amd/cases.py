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


def _grid_metrics(st: PomState, delx: float, stretch: float) -> None:
    im, jm = st.im, st.jm
    i = np.arange(1, im + 1)[None, :]
    j = np.arange(1, jm + 1)[:, None]
    st.dx[...] = delx - stretch * delx * np.sin(math.pi * i / im) / 2.0 + 0.0 * j
    st.dy[...] = delx - stretch * delx * np.sin(math.pi * j / jm) / 2.0 + 0.0 * i
    # corner / centre coordinates by accumulating spacings
    ec = np.zeros((jm, im))
    nc = np.zeros((jm, im))
    ec[:, 1:] = np.cumsum(st.dx[:, :-1], axis=1)
    nc[1:, :] = np.cumsum(st.dy[:-1, :], axis=0)
    st.east_c[...] = ec
    st.north_c[...] = nc
    st.east_e[...] = ec + 0.5 * st.dx
    st.north_e[...] = nc + 0.5 * st.dy
    st.east_u[...] = ec
    st.north_u[...] = nc + 0.5 * st.dy
    st.east_v[...] = ec + 0.5 * st.dx
    st.north_v[...] = nc
    st.rot[...] = 0.0


def _derive_grid(st: PomState, lat0: float, lat_span: float) -> None:
    """What read_grid derives after the file read (initialize.f:329-386) + the u/v masks."""
    kb = st.kb
    st.dz[:kb - 1] = st.z[:kb - 1] - st.z[1:]
    st.dzz[:kb - 1] = st.zz[:kb - 1] - st.zz[1:]
    st.dz[kb - 1] = 0.0
    st.dzz[kb - 1] = 0.0
    ymax = float(st.north_e.max())
    lat = lat0 + lat_span * st.north_e / ymax
    st.cor[...] = 2.0 * 7.29e-5 * np.sin(lat * (st.pi / 180.0))
    st.period = (2.0 * st.pi) / abs(st.cor[st.jm // 2 - 1, st.im // 2 - 1]) / 86400.0
    st.art[...] = st.dx * st.dy
    st.aru[1:, 1:] = 0.25 * (st.dx[1:, 1:] + st.dx[1:, :-1]) * (st.dy[1:, 1:] + st.dy[1:, :-1])
    st.arv[1:, 1:] = 0.25 * (st.dx[1:, 1:] + st.dx[:-1, 1:]) * (st.dy[1:, 1:] + st.dy[:-1, 1:])
    st.aru[:, 0] = st.aru[:, 1]
    st.arv[:, 0] = st.arv[:, 1]
    st.aru[0, :] = st.aru[1, :]
    st.arv[0, :] = st.arv[1, :]
    # masks (io_pnetcdf.F:2243-2256)
    fsm = st.fsm
    st.dvm[...] = fsm
    st.dum[...] = fsm
    st.dvm[1:, :][(fsm[:-1, :] == 0.0) & (fsm[1:, :] != 0.0)] = 0.0
    st.dum[:, 1:][(fsm[:, :-1] == 0.0) & (fsm[:, 1:] != 0.0)] = 0.0
    st.d[...] = st.h + st.el
    st.dt[...] = st.h + st.et


def make_case(name: str, im: int, jm: int, kb: int, **nml) -> PomState:
    """Global (single-tile) state holding everything the reference's readers + read_grid provide."""
    st = PomState(im, jm, kb)
    c = run_constants(None, **nml)
    apply_constants(st, c)
    z, zz = sigma_levels(kb)
    st.z[...] = z
    st.zz[...] = zz
    delx = 8000.0
    _grid_metrics(st, delx, 1.0 if name != "basin" else 0.5)
    xc = st.east_e - st.east_e[:, (im + 1) // 2 - 1][:, None]
    yc = st.north_e - st.north_e[(jm + 1) // 2 - 1, :][None, :]
    lx = float(st.east_e.max())
    ly = float(st.north_e.max())
    vel = 0.0
    if name == "seamount":
        ra = 0.12 * min(lx, ly) + 12500.0
        st.h[...] = 4500.0 * (1.0 - 0.9 * np.exp(-(xc * xc + yc * yc) / (ra * ra)))
        st.h[st.h < 1.0] = 1.0
        st.h[0, :] = 1.0
        st.h[jm - 1, :] = 1.0
        vel = 0.2
    elif name == "island":
        ra = 0.10 * min(lx, ly) + 10000.0
        st.h[...] = 3000.0 * (1.0 - 0.85 * np.exp(-(xc * xc + 1.7 * yc * yc) / (ra * ra)))
        st.h[st.h < 1.0] = 1.0
        vel = 0.1
    elif name == "basin":
        sx = st.east_e / lx
        sy = st.north_e / ly
        st.h[...] = (1500.0 + 1500.0 * sx + 400.0 * np.sin(3.0 * math.pi * sx) * np.sin(2.0 * math.pi * sy)
                     + 250.0 * np.cos(7.0 * math.pi * sy))
        st.h[0, :] = 1.0
        st.h[jm - 1, :] = 1.0
        st.h[:, 0] = 1.0
        st.h[:, im - 1] = 1.0
    else:
        raise ValueError(f"unknown case {name!r}")
    st.fsm[...] = np.where(st.h > 1.0, 1.0, 0.0)
    _derive_grid(st, 30.0, 15.0)

    # initial T/S and climatology (level kb defined explicitly -- the reference's reader leaves it
    # uninitialised, io_pnetcdf.F:2825-2839)
    depth = zz[:, None, None] * st.h[None, :, :]
    if name == "basin":
        rng = np.random.default_rng(12345)
        st.tb[...] = 18.0 + 14.0 * depth / 3500.0 + 1.0e-3 * rng.standard_normal(st.tb.shape)
        st.sb[...] = 35.0 - 0.6 * depth / 3500.0
        st.tclim[...] = 18.0 + 14.0 * depth / 3500.0
        st.sclim[...] = st.sb
    else:
        st.tb[...] = 5.0 + 15.0 * np.exp(depth / 1000.0) - st.tbias
        st.sb[...] = 35.0 - st.sbias
        st.tclim[...] = st.tb
        st.sclim[...] = st.sb
    st.tb[kb - 1] = st.tb[kb - 2]
    st.sb[kb - 1] = st.sb[kb - 2]
    st.tclim[kb - 1] = st.tclim[kb - 2]
    st.sclim[kb - 1] = st.sclim[kb - 2]
    st.tb[...] *= st.fsm[None]
    st.sb[...] *= st.fsm[None]
    st.tclim[...] *= st.fsm[None]
    st.sclim[...] *= st.fsm[None]

    # initial flow and open-boundary data
    st.ub[:kb - 1] = vel * st.dum[None]
    st.uab[...] = vel * st.dum
    st.uabe[...] = st.uab[:, im - 2]
    st.uabw[...] = st.uab[:, 1]
    if name == "island":
        st.vb[:kb - 1] = 0.25 * vel * st.dvm[None]
        st.vab[...] = 0.25 * vel * st.dvm
        st.vabn[...] = st.vab[jm - 2, :]
        st.vabs[...] = st.vab[1, :]
        st.vabe[...] = st.vab[:, im - 1]
        st.vabw[...] = st.vab[:, 0]
        st.uabn[...] = st.uab[jm - 1, :]
        st.uabs[...] = st.uab[0, :]
        st.vbn[...] = st.vb[:, jm - 2, :]
        st.vbs[...] = st.vb[:, 1, :]
        st.ubn[...] = st.ub[:, jm - 1, :]
        st.ubs[...] = st.ub[:, 0, :]
    st.ube[...] = st.ub[:, :, im - 2]
    st.ubw[...] = st.ub[:, :, 1]

    # surface forcing, constant in time (the reference's wind/heat readers are out of scope)
    if name == "basin":
        st.wusurf[...] = -1.0e-4 * np.cos(math.pi * st.north_e / ly) * st.fsm
        st.wtsurf[...] = 2.0e-6 * np.sin(2.0 * math.pi * st.east_e / lx) * st.fsm
    else:
        st.wusurf[...] = -2.0e-5 * st.fsm
        st.wvsurf[...] = 1.0e-5 * np.sin(math.pi * st.east_e / lx) * st.fsm

    # relaxation targets served to restore_interior (bounds_forcing.f:1039-1065 reads record
    # iint/irst+1 at iint=2 and the following one): climatology, the second record 0.01 degC /
    # 0.002 psu warmer/saltier so that the time interpolation is exercised.
    st.restore_records = [
        (st.tclim[:, :jm, :im].copy(), st.sclim[:, :jm, :im].copy()),
        ((st.tclim[:, :jm, :im] + 0.01) * st.fsm[None, :jm, :im], (st.sclim[:, :jm, :im] + 0.002) * st.fsm[None, :jm, :im]),
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
    if hasattr(g, "restore_records"):
        st.restore_records = [(np.ascontiguousarray(a[:, j0:j0 + jm, i0:i0 + im]),
                               np.ascontiguousarray(b[:, j0:j0 + jm, i0:i0 + im])) for a, b in g.restore_records]
    return st
