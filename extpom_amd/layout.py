"""Host-side mirror of the reference's COMMON-block storage (reference pom.h_dist:46-640).

The member order of every block is read from ``include/pom_layout.h`` -- the one place the
layout contract is written down -- so Python, C, HIP and the generated Fortran include agree
by construction.  ``PomState`` owns one contiguous float64 buffer per block, laid out exactly
like the Fortran COMMON block, and exposes every member as a numpy view:

    st.u[k-1, j-1, i-1]   <->   u(i,j,k)     (column-major, i contiguous)

so a block can be handed to the C ABI (or compared with the reference's own block) by address.
"""
from __future__ import annotations

import os
import re

import numpy as np

_HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "pom_layout.h")


def _macro_body(text: str, name: str) -> str:
    m = re.search(r"#define\s+" + name + r"\([^)]*\)\s*\\\n((?:.*\\\n)*.*)\n", text)
    if not m:
        raise RuntimeError(f"{name} not found in pom_layout.h")
    return m.group(1).replace("\\\n", " ")


def _parse():
    text = open(_HDR).read()
    one = lambda n: re.findall(r"X\((\w+)\)", _macro_body(text, n))
    blk1d, blk2d, blk3d, blksiz = one("POM_BLK1D"), one("POM_BLK2D"), one("POM_BLK3D"), one("POM_BLKSIZ")
    bdry = re.findall(r"X\((\w+),(\w+)\)", _macro_body(text, "POM_BDRY"))
    con = re.findall(r"\b([DI])\((\w+)\)", _macro_body(text, "POM_BLKCON"))
    return blk1d, blk2d, blk3d, bdry, con, blksiz


BLK1D, BLK2D, BLK3D, BDRY, BLKCON, BLKSIZ = _parse()
P1 = {n: i for i, n in enumerate(BLK1D)}
P2 = {n: i for i, n in enumerate(BLK2D)}
P3 = {n: i for i, n in enumerate(BLK3D)}

# blkcon: 22 doubles, 4 int32, 16 doubles, 14 int32 (pom.h_dist:142-198) -- 376 bytes, no padding
CON_DTYPE = np.dtype([(n, "<f8" if k == "D" else "<i4") for k, n in BLKCON])
assert CON_DTYPE.itemsize == 376
SIZ_DTYPE = np.dtype([(n, "<i4") for n in BLKSIZ])
assert SIZ_DTYPE.itemsize == 32

# fields written by the reference's restart writer (io_pnetcdf.F:1724-1886) = the prognostic state
RESTART_2D = ["wubot", "wvbot", "aam2d", "ua", "uab", "va", "vab", "el", "elb", "et", "etb", "egb",
              "utb", "vtb", "adx2d", "ady2d", "advua", "advva"]
RESTART_3D = ["u", "ub", "v", "vb", "w", "t", "tb", "s", "sb", "rho", "km", "kh", "kq", "l", "q2",
              "q2b", "aam", "q2l", "q2lb"]
# the five fields north_star names for the parity bar
PROGNOSTIC = ["el", "et", "ua", "va", "u", "v", "t", "s"]


def bdry_shape(kind: str, iml: int, jml: int, kb: int):
    return {"J": (jml,), "I": (iml,), "JK": (kb, jml), "IK": (kb, iml)}[kind]


class PomState:
    """All COMMON-block state of one tile, in the reference's storage layout."""

    def __init__(self, im_local: int, jm_local: int, kb: int, im: int | None = None, jm: int | None = None):
        self.__dict__["_views"] = {}
        self.im_local, self.jm_local, self.kb = int(im_local), int(jm_local), int(kb)
        self.im = int(im if im is not None else im_local)
        self.jm = int(jm if jm is not None else jm_local)
        n2 = self.im_local * self.jm_local
        self.blk1d = np.zeros((len(BLK1D), kb))
        self.blk2d = np.zeros((len(BLK2D), self.jm_local, self.im_local))
        self.blk3d = np.zeros((len(BLK3D), kb, self.jm_local, self.im_local))
        nb = sum(int(np.prod(bdry_shape(k, im_local, jm_local, kb))) for _, k in BDRY)
        self.bdry = np.zeros(nb)
        self.con = np.zeros(1, dtype=CON_DTYPE)
        # neighbours as in blkpar (parallel_mpi.f:111-119): -1 = physical edge
        self.n_west = self.n_east = self.n_south = self.n_north = -1
        self.i_off = 0  # global i of local i is i + i_off (parallel_mpi.f:82)
        self.j_off = 0
        v = self._views
        for n, i in P1.items():
            v[n] = self.blk1d[i]
        for n, i in P2.items():
            v[n] = self.blk2d[i]
        for n, i in P3.items():
            v[n] = self.blk3d[i]
        off = 0
        for n, k in BDRY:
            shp = bdry_shape(k, im_local, jm_local, kb)
            cnt = int(np.prod(shp))
            v[n] = self.bdry[off:off + cnt].reshape(shp)
            off += cnt
        del n2

    # -- attribute access: arrays are views, scalars live in the blkcon record -------------
    def __getattr__(self, name):
        v = self.__dict__["_views"]
        if name in v:
            return v[name]
        if name in CON_DTYPE.names:
            return self.__dict__["con"][name][0].item()
        raise AttributeError(name)

    def __setattr__(self, name, value):
        v = self.__dict__["_views"]
        if name in v:
            v[name][...] = value
        elif "con" in self.__dict__ and name in CON_DTYPE.names:
            self.__dict__["con"][name][0] = value
        else:
            self.__dict__[name] = value

    @property
    def siz(self) -> np.ndarray:
        s = np.zeros(1, dtype=SIZ_DTYPE)
        s["im"], s["imm1"], s["imm2"] = self.im, self.im - 1, self.im - 2
        s["jm"], s["jmm1"], s["jmm2"] = self.jm, self.jm - 1, self.jm - 2
        s["kbm1"], s["kbm2"] = self.kb - 1, self.kb - 2
        return s

    def copy(self) -> "PomState":
        o = PomState(self.im_local, self.jm_local, self.kb, self.im, self.jm)
        o.blk1d[...] = self.blk1d
        o.blk2d[...] = self.blk2d
        o.blk3d[...] = self.blk3d
        o.bdry[...] = self.bdry
        o.con[...] = self.con
        for a in ("n_west", "n_east", "n_south", "n_north", "i_off", "j_off"):
            setattr(o, a, getattr(self, a))
        if "restore_records" in self.__dict__:
            o.restore_records = self.restore_records
        if "lateral_records" in self.__dict__:
            o.lateral_records = self.lateral_records
        if "forcing_records" in self.__dict__:
            o.forcing_records = self.forcing_records
        if "lramp" in self.__dict__:
            o.lramp = self.lramp
        return o

    def field(self, name: str) -> np.ndarray:
        return self._views[name]
