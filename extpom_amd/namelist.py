"""``pom.nml`` reader and run constants: the behaviour of the reference's ``read_input``
(reference pom/initialize.f:67-244).

Hard-coded physics constants first (initialize.f:80-168), then the namelist group ``pom_nml``
overrides its members (initialize.f:71-74,173-175), then the derived time-step constants
(initialize.f:178-198).  Unknown keys are an error, as a Fortran namelist read would make them.
"""
from __future__ import annotations

import math
import re

import numpy as np

# members of namelist /pom_nml/ (initialize.f:71-74) with the Python type they are read as
NML_KEYS = {
    "title": str, "wrk_pth": str, "netcdf_file": str, "mode": int, "nadv": int, "nitera": int,
    "sw": float, "npg": int, "dte": float, "isplit": int, "time_start": str, "nread_rst": int,
    "read_rst_file": str, "cont_bry": int, "write_rst": float, "write_rst_file": str,
    "days": float, "prtd1": float, "prtd2": float, "swtch": float, "ntp": int, "nbct": int,
    "nbcs": int,
}

# constants set before the namelist is read (initialize.f:80-168)
DEFAULTS = dict(
    lramp=False, rhoref=1025.0, tbias=0.0, sbias=0.0, grav=9.806, kappa=0.4, z0b=0.01,
    cbcmin=0.0025, cbcmax=1.0, horcon=0.1, tprni=0.1, umol=1.0e-6, vmaxl=100.0, slmax=2.0,
    ntp=2, nbct=1, nbcs=1, ispadv=1, smoth=0.10, alpha=0.0, aam_init=0.0,
)

# values a run must define through pom.nml; these are pom.nml_dist:2-21's
NML_DIST = dict(
    title="run", wrk_pth="./", netcdf_file="nonetcdf", mode=3, nadv=2, nitera=1, sw=0.5, npg=1,
    dte=2.0, isplit=30, time_start="2000-01-01 00:00:00 +00:00", nread_rst=0,
    read_rst_file="restart.0001.nc", cont_bry=0, write_rst=1.0, write_rst_file="restart",
    days=1.0, prtd1=0.1, prtd2=1.0, swtch=9999.0,
)


def _fortran_value(tok: str, typ):
    tok = tok.strip().rstrip(",")
    if typ is str:
        if len(tok) >= 2 and tok[0] in "'\"" and tok[-1] == tok[0]:
            return tok[1:-1]
        return tok
    if typ is int:
        return int(float(tok.lower().replace("d", "e")))
    return float(tok.lower().replace("d", "e"))


def parse_namelist(text: str) -> dict:
    """Parse the ``&pom_nml ... /`` group of a namelist file."""
    m = re.search(r"&pom_nml(.*?)^\s*/", text, flags=re.S | re.M | re.I)
    if not m:
        raise ValueError("namelist group &pom_nml not found")
    out = {}
    for line in m.group(1).splitlines():
        line = line.split("!")[0].strip() if "'" not in line else re.sub(r"!(?=(?:[^']*'[^']*')*[^']*$).*", "", line).strip()
        if not line:
            continue
        if "=" not in line:
            raise ValueError(f"bad namelist line: {line!r}")
        key, val = line.split("=", 1)
        key = key.strip().lower()
        if key not in NML_KEYS:
            raise ValueError(f"'{key}' is not a member of namelist pom_nml")
        out[key] = _fortran_value(val, NML_KEYS[key])
    return out


def nint(x: float) -> int:
    """Fortran NINT: round half away from zero."""
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def run_constants(nml: dict | None = None, **overrides) -> dict:
    """Everything ``read_input`` leaves in COMMON /blkcon/ (+ ``lramp``), as a dict."""
    c = dict(DEFAULTS)
    c.update(NML_DIST)
    if nml:
        c.update(nml)
    c.update(overrides)
    c["small"] = 1.0e-9                                   # initialize.f:179
    c["pi"] = math.atan(1.0) * 4.0                        # initialize.f:180
    isplit = int(c["isplit"])
    # dti=dte*float(isplit): FLOAT gives REAL(4), exact for any sensible isplit (initialize.f:182)
    c["dti"] = c["dte"] * float(np.float32(isplit))
    c["dte2"] = c["dte"] * 2
    c["dti2"] = c["dti"] * 2
    c["iend"] = max(nint(c["days"] * 24.0 * 3600.0 / c["dti"]), 2)
    c["iprint"] = nint(c["prtd1"] * 24.0 * 3600.0 / c["dti"])
    c["iswtch"] = nint(c["swtch"] * 24.0 * 3600.0 / c["dti"])
    c["irestart"] = nint(c["write_rst"] * 24.0 * 3600.0 / c["dti"])
    c["ispi"] = 1.0 / float(np.float32(isplit))           # initialize.f:192
    c["isp2i"] = 1.0 / (2.0 * float(np.float32(isplit)))  # initialize.f:193
    c["time0"] = 0.0
    c["time"] = 0.0
    if c["nread_rst"] == 0:
        c["cont_bry"] = 0
    c.setdefault("ramp", 1.0)
    c.setdefault("period", 0.0)
    c.setdefault("rfe", 1.0)
    c.setdefault("rfw", 1.0)
    c.setdefault("rfn", 1.0)
    c.setdefault("rfs", 1.0)
    c["iint"] = 0
    c["iext"] = 0
    c["error_status"] = 0
    return c


def read_namelist_file(path: str, **overrides) -> dict:
    with open(path) as f:
        return run_constants(parse_namelist(f.read()), **overrides)


def apply_constants(st, c: dict) -> None:
    """Store run constants into a PomState's blkcon record."""
    from .layout import CON_DTYPE
    for name in CON_DTYPE.names:
        if name in c:
            st.con[name][0] = c[name]
    st.lramp = bool(c.get("lramp", False))
