"""TEST INFRASTRUCTURE ONLY -- ctypes driver for the reference build in ``oracle/_ref``.

``oracle/build_ref.sh`` compiles the *unmodified* reference hot path (solver.f advance.f
bounds_forcing.f initialize.f parallel_mpi.f) into ``oracle/_ref/libpomref_<im>x<jm>x<kb>.so``.
This module copies a ``PomState`` into that library's COMMON blocks, calls the reference's own
subroutines by their Fortran symbol names, and copies the blocks back.  It is used (i) to pin
the C restatement in ``oracle/pom_oracle.c`` and (ii) to generate the golden fixtures committed
under ``tests/golden``.  Nothing in the product imports it, and it only works where
``/root/reference`` was present at build time (this container; never the GPU box's product path).

``advance()`` restates the ten-line sequencing of the reference's ``advance`` (advance.f:6-59)
without its file-driven forcing, print and output calls, which need PnetCDF input files.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def ref_path(im, jm, kb, im_local=None, jm_local=None, n_proc=1):
    tag = f"{im}x{jm}x{kb}"
    if n_proc != 1:
        tag += f"_{im_local}x{jm_local}p{n_proc}"
    return os.path.join(HERE, "_ref", f"libpomref_{tag}.so")


def have_ref(im, jm, kb) -> bool:
    return os.path.exists(ref_path(im, jm, kb))


class RefLib:
    def __init__(self, im_local, jm_local, kb, path=None):
        self.iml, self.jml, self.kb = im_local, jm_local, kb
        path = path or ref_path(im_local, jm_local, kb)
        self.lib = ctypes.CDLL(path)
        self._frecs, self._lrecs = [], []
        n2 = im_local * jm_local
        from extpom_amd.layout import BLK1D, BLK2D, BLK3D, CON_DTYPE, SIZ_DTYPE
        self._n = dict(blk1d=len(BLK1D) * kb, blk2d=len(BLK2D) * n2, blk3d=len(BLK3D) * n2 * kb)

        def dbl(sym, n):
            return np.ctypeslib.as_array((ctypes.c_double * n).in_dll(self.lib, sym))

        self.c1 = dbl("blk1d_", self._n["blk1d"])
        self.c2 = dbl("blk2d_", self._n["blk2d"])
        self.c3 = dbl("blk3d_", self._n["blk3d"])
        self.con = np.ctypeslib.as_array((ctypes.c_char * CON_DTYPE.itemsize).in_dll(self.lib, "blkcon_")).view(CON_DTYPE)
        self.siz = np.ctypeslib.as_array((ctypes.c_char * SIZ_DTYPE.itemsize).in_dll(self.lib, "blksiz_")).view(SIZ_DTYPE)
        npar = 3 + im_local + jm_local + 4
        self.par = np.ctypeslib.as_array((ctypes.c_int * npar).in_dll(self.lib, "blkpar_"))
        self.log = np.ctypeslib.as_array((ctypes.c_int * 1).in_dll(self.lib, "blklog_"))
        self._bd = None

    def _bdry(self, n):
        if self._bd is None:
            self._bd = np.ctypeslib.as_array((ctypes.c_double * n).in_dll(self.lib, "bdry_"))
        return self._bd

    # ---- state transfer ------------------------------------------------------------------
    def put(self, st):
        assert (st.im_local, st.jm_local, st.kb) == (self.iml, self.jml, self.kb)
        self.c1[:] = st.blk1d.ravel()
        self.c2[:] = st.blk2d.ravel()
        self.c3[:] = st.blk3d.ravel()
        self._bdry(st.bdry.size)[:] = st.bdry
        self.con[:] = st.con
        self.siz[:] = st.siz
        p = self.par
        p[0], p[1], p[2] = 0, 0, 0
        p[3:3 + self.iml] = np.where(np.arange(1, self.iml + 1) <= st.im, np.arange(1, self.iml + 1) + st.i_off, 0)
        p[3 + self.iml:3 + self.iml + self.jml] = np.where(np.arange(1, self.jml + 1) <= st.jm,
                                                          np.arange(1, self.jml + 1) + st.j_off, 0)
        p[-4:] = [st.n_west, st.n_east, st.n_south, st.n_north]
        self.log[0] = 1 if getattr(st, "lramp", False) else 0
        # relaxation targets for restore_interior: see ref_traps.c (input hook, no arithmetic)
        self._recs = []
        for n, (tr, sr) in enumerate(getattr(st, "restore_records", []), start=1):
            tr = np.ascontiguousarray(tr, dtype=np.float64)
            sr = np.ascontiguousarray(sr, dtype=np.float64)
            self._recs.append((tr, sr))
            self.lib.pomref_set_restore_record(ctypes.c_int(n), ctypes.c_void_p(tr.ctypes.data),
                                               ctypes.c_void_p(sr.ctypes.data), ctypes.c_size_t(tr.size))

        # surface-forcing records (wind / heat / surface), see ref_traps.c: input hooks as well
        self._frecs = []
        for kind, name in enumerate(("wind", "heat", "surface")):
            for n, (a, b) in enumerate(getattr(st, "forcing_records", {}).get(name, []), start=1):
                a = np.ascontiguousarray(a, dtype=np.float64)
                b = np.ascontiguousarray(b, dtype=np.float64)
                self._frecs.append((a, b))
                self.lib.pomref_set_forcing_record(ctypes.c_int(kind), ctypes.c_int(n), ctypes.c_void_p(a.ctypes.data),
                                                   ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(a.size))

        # lateral boundary records (20 arrays each), see ref_traps.c
        self._lrecs = []
        for n, rec in enumerate(getattr(st, "lateral_records", []), start=1):
            arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in rec]
            ptrs = (ctypes.c_void_p * 20)(*[a.ctypes.data for a in arrs])
            cnts = (ctypes.c_size_t * 20)(*[a.size for a in arrs])
            self._lrecs.append((arrs, ptrs, cnts))
            self.lib.pomref_set_lateral_record(ctypes.c_int(n), ptrs, cnts)

    def get(self, st):
        st.blk1d[...] = self.c1.reshape(st.blk1d.shape)
        st.blk2d[...] = self.c2.reshape(st.blk2d.shape)
        st.blk3d[...] = self.c3.reshape(st.blk3d.shape)
        st.bdry[...] = self._bdry(st.bdry.size)
        st.con[...] = self.con
        return st

    # ---- calls ---------------------------------------------------------------------------
    def f3(self, name):
        """Address of a 3-D COMMON array inside the library (for routines with array arguments)."""
        from extpom_amd.layout import P3
        n3 = self.iml * self.jml * self.kb
        return ctypes.c_void_p(self.c3.ctypes.data + 8 * P3[name] * n3)

    def f2(self, name):
        from extpom_amd.layout import P2
        return ctypes.c_void_p(self.c2.ctypes.data + 8 * P2[name] * self.iml * self.jml)

    def call(self, name, *args):
        fn = getattr(self.lib, name + "_")
        fn.restype = None
        fn(*args)

    def mpi_init(self):
        """the reference's own initialize_mpi (parallel_mpi.f:6-20) once per process: a singleton MPI_COMM_WORLD,
        needed only by routines that reduce over ranks (domain_stats -> sum0d_mpi)"""
        flag = ctypes.c_int(0)
        self.lib.MPI_Initialized(ctypes.byref(flag))
        if not flag.value:
            keep = int(self.con["error_status"][0])
            self.call("initialize_mpi")
            self.con["error_status"][0] = keep

    def call_idx(self, name, idx):
        self.call(name, ctypes.byref(ctypes.c_int(idx)))

    def advance(self):
        """One internal step: hot-path sequence of advance.f:6-59 (iint already set by the caller)."""
        self.call("get_time")
        if self._frecs:                     # advance.f:14-18; the readers are input hooks (ref_traps.c)
            self.call("surface_forcing")
        if self._lrecs:
            self.call("lateral_bc")
        self.call("lateral_viscosity")
        self.call("mode_interaction")
        isplit = int(self.con["isplit"][0])
        for iext in range(1, isplit + 1):
            self.con["iext"][0] = iext
            self.call("mode_external")
        # a Fortran DO variable is left at last+1 (advance.f:27-29)
        self.con["iext"][0] = isplit + 1
        self.call("mode_internal")
        self.call("check_velocity")

    def run(self, st, nsteps, iint0=0):
        self.put(st)
        for n in range(1, nsteps + 1):
            self.con["iint"][0] = iint0 + n
            self.advance()
        return self.get(st)


# --- the dens / baropg callbacks finish_initial() expects, served by the reference itself ------
def ref_finish_initial(st):
    from extpom_amd.cases import finish_initial
    lib = RefLib(st.im_local, st.jm_local, st.kb)

    def dens(s, si, ti, rho):
        lib.put(s)
        lib.call("dens", lib.f3(si), lib.f3(ti), lib.f3(rho))
        lib.get(s)

    def baropg(s):
        lib.put(s)
        lib.call("baropg_mcc" if int(s.npg) == 2 else "baropg")
        lib.get(s)

    return finish_initial(st, dens, baropg)
