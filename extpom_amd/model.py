"""Host-side mirror of the reference's program flow for the hot path, over the C ABI.

``PomGpu`` owns one tile's device context.  Method names are the reference's subroutine names
(advance, lateral_viscosity, mode_interaction, mode_external, mode_internal, check_velocity, advct,
advq, ... : reference pom/advance.f, pom/solver.f); array arguments are given by field NAME and
resolved to the host address of that COMMON array -- what a Fortran caller passes by reference.
State crosses the boundary only through ``upload`` / ``download`` (whole COMMON blocks).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import lib as _lib
from .layout import P2, P3, PomState
from .lib import Dims, PomGpuError


class PomGpu:
    def __init__(self, st: PomState, device: int = 0, stream: int | None = None, libpath: str | None = None):
        self.L = _lib.load(libpath)
        self.st = st
        d = Dims(st.im, st.jm, st.kb, st.im_local, st.jm_local, st.n_west, st.n_east, st.n_south, st.n_north)
        h = ctypes.c_void_p()
        rc = self.L.pomgpu_create(ctypes.byref(h), ctypes.byref(d), device, ctypes.c_void_p(stream or 0))
        if rc != 0:
            raise PomGpuError(f"pomgpu_create failed with status {rc} (no CPU fallback exists for the hot path)")
        self.h = h
        self._exch_cb = None
        self.upload()

    # ---- plumbing ------------------------------------------------------------------------
    def _chk(self, rc, what):
        if rc != 0:
            raise PomGpuError(f"{what}: status {rc}: {self.L.pomgpu_last_error(self.h).decode()}")

    @staticmethod
    def _p(a):
        return ctypes.c_void_p(a.ctypes.data)

    def tune_placement(self, steps: int = 3, max_try: int = 6):
        """pomgpu_tune_placement: try a few start offsets of the 3-D arrays inside one allocation, keep the fastest (the model
        advances by tried x (steps + 1) internal steps).  Returns {"tried": n, "front_mib": [...], "ms_per_step": [...], "kept": k}."""
        ms = (ctypes.c_double * max_try)()
        fr = (ctypes.c_long * max_try)()
        pd = (ctypes.c_long * max_try)()
        n, k = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(self.L.pomgpu_tune_placement(self.h, steps, max_try, ms, fr, pd, ctypes.byref(n), ctypes.byref(k)), "tune_placement")
        return {"tried": n.value, "front_mib": [int(fr[q]) for q in range(n.value)], "pad_mib": [int(pd[q]) for q in range(n.value)],
                "ms_per_step": [round(float(ms[q]), 3) for q in range(n.value)], "kept": k.value}

    def close(self):
        if getattr(self, "h", None):
            self.L.pomgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, st: PomState | None = None):
        st = st or self.st
        self.st = st
        self._chk(self.L.pomgpu_upload(self.h, self._p(st.blk1d), self._p(st.blk2d), self._p(st.blk3d), self._p(st.bdry),
                                       self._p(st.con), 1 if getattr(st, "lramp", False) else 0), "upload")
        self._chk(self.L.pomgpu_bind_host(self.h, self._p(st.blk2d), self._p(st.blk3d)), "bind_host")
        for n, (tr, sr) in enumerate(getattr(st, "restore_records", []), start=1):
            tr = np.ascontiguousarray(tr, dtype=np.float64)
            sr = np.ascontiguousarray(sr, dtype=np.float64)
            self._chk(self.L.pomgpu_set_restore_record(self.h, n, self._p(tr), self._p(sr)), "set_restore_record")

    def set_forcing_records(self, first: int = 1, count: int = 4):
        """hand records first..first+count-1 of st.forcing_records (wind / heat / surface) to the library: what the
        host's read_wind_pnetcdf etc. would deliver; the library keeps the last four per kind"""
        for kind, name in enumerate(("wind", "heat", "surface")):
            recs = getattr(self.st, "forcing_records", {}).get(name, [])
            for n in range(first, min(first + count, len(recs) + 1)):
                a = np.ascontiguousarray(recs[n - 1][0], dtype=np.float64)
                b = np.ascontiguousarray(recs[n - 1][1], dtype=np.float64)
                self._chk(self.L.pomgpu_set_forcing_record(self.h, kind, n, self._p(a), self._p(b)), "set_forcing_record")

    def set_lateral_records(self, first: int = 1, count: int = 4):
        """hand records first..first+count-1 of st.lateral_records (20 arrays each, see pomgpu.h) to the library"""
        recs = getattr(self.st, "lateral_records", [])
        for n in range(first, min(first + count, len(recs) + 1)):
            arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in recs[n - 1]]
            ptrs = (ctypes.c_void_p * 20)(*[a.ctypes.data for a in arrs])
            self._chk(self.L.pomgpu_set_lateral_record(self.h, n, ptrs), "set_lateral_record")

    def download(self, st: PomState | None = None) -> PomState:
        st = st or self.st
        self._chk(self.L.pomgpu_download(self.h, self._p(st.blk1d), self._p(st.blk2d), self._p(st.blk3d), self._p(st.bdry),
                                         self._p(st.con)), "download")
        return st

    def set_con(self, **kw):
        """update blkcon scalars (iint, iext, ...) on the library side"""
        for k, v in kw.items():
            self.st.con[k][0] = v
        self._chk(self.L.pomgpu_set_con(self.h, self._p(self.st.con), 1 if getattr(self.st, "lramp", False) else 0),
                  "set_con")

    def get_con(self):
        self._chk(self.L.pomgpu_get_con(self.h, self._p(self.st.con)), "get_con")
        return self.st.con

    def sync(self):
        self._chk(self.L.pomgpu_sync(self.h), "sync")

    def current_stream(self) -> int:
        """the hipStream_t the library is enqueueing on right now: inside a transport callback, the stream of the round being served"""
        return int(self.L.pomgpu_current_stream(self.h) or 0)

    def device_ptr(self, name: str) -> int:
        if name in P3:
            return self.L.pomgpu_device_3d(self.h, P3[name])
        return self.L.pomgpu_device_2d(self.h, P2[name])

    def set_exchange(self, fn):
        """fn(list_of_device_addresses, list_of_levels) -- see extpom_amd.halo"""
        def cb(user, ptrs, nz, count):
            fn([ptrs[n] for n in range(count)], [nz[n] for n in range(count)])
        self._exch_cb = _lib.EXCHANGE_FN(cb)
        self._chk(self.L.pomgpu_set_exchange(self.h, self._exch_cb, None), "set_exchange")

    # ---- the library's own exchange (pomgpu.h "transport") ------------------------------------
    @staticmethod
    def neighbours8(tile):
        """ranks in the C ABI's direction order W E S N SW SE NW NE"""
        return [tile.n_west, tile.n_east, tile.n_south, tile.n_north, tile.n_sw, tile.n_se, tile.n_nw, tile.n_ne]

    def set_transport(self, tile, fn, agree=None, stream_ordered=False):
        """callback mover (tests): fn(send, scount, recv, rcount), each a list of eight (device address, doubles).
        stream_ordered: fn enqueues its copies on self.current_stream() (pomgpu_transport_stream_ordered): rounds of the library's second
        stream reach it without that stream having been completed first.
        agree: the host's reduction over ALL ranks -- agree(mine: int) -> min over the ranks -- through which the ranks
        settle whether message rounds may run on the library's second stream (all of them or none, pomgpu.h); without
        it every round stays on the main stream."""
        def cb(user, send, scount, recv, rcount):
            fn([send[d] for d in range(8)], [scount[d] for d in range(8)], [recv[d] for d in range(8)], [rcount[d] for d in range(8)])
        self._tp_cb = _lib.TRANSPORT_FN(cb)
        nb = (ctypes.c_int * 8)(*self.neighbours8(tile))
        self._chk(self.L.pomgpu_set_transport(self.h, nb, self._tp_cb, None), "set_transport")
        if stream_ordered:
            self._chk(self.L.pomgpu_transport_stream_ordered(self.h, 1), "transport_stream_ordered")
        if agree is not None:
            self.side_agree(agree)

    def side_capable(self) -> int:
        """this rank's own answer: could it serve message rounds on a second stream?"""
        return int(self.L.pomgpu_transport_side_capable(self.h))

    def side_agree(self, agree):
        """collective: agree(mine) must return the minimum of `mine` over all ranks of the decomposition.  The ranks first
        compare the digest of the switches that choose which message rounds exist (pomgpu.h): ranks started with different
        POMGPU_* sets are all refused here instead of hanging in a later round."""
        d = int(self.L.pomgpu_switch_digest(self.h))
        lo, hi = int(agree(d)), -int(agree(-d))
        if lo != hi:
            raise PomGpuError("the ranks were created under different POMGPU_* switch sets (those that choose the message rounds): "
                              "give every rank the same environment")
        self._chk(self.L.pomgpu_transport_side_agree(self.h, int(agree(self.side_capable()))), "transport_side_agree")

    def switch(self, name: str, value=None):
        """one developer switch of this live context (pomgpu_debug_switch): value None unsets it.  The library reads its
        switches from the environment once, when the context is created."""
        v = None if value is None else str(value).encode()
        self._chk(self.L.pomgpu_debug_switch(self.h, name.encode(), v), "debug_switch")

    def rccl_nranks(self) -> int:
        """what ncclCommCount says about the transport's communicator (0: no RCCL transport)"""
        return int(self.L.pomgpu_rccl_nranks(self.h))

    def rccl_init(self, tile, id128: bytes, rank: int, nranks: int, librccl: str | None = None):
        """production mover: grouped ncclSend / ncclRecv on the library's stream"""
        nb = (ctypes.c_int * 8)(*self.neighbours8(tile))
        buf = ctypes.create_string_buffer(bytes(id128), 128)
        self._chk(self.L.pomgpu_rccl_init(self.h, buf, rank, nranks, nb, librccl.encode() if librccl else None), "rccl_init")

    def clear_transport(self):
        """back to a single tile's behaviour (or to the hooks installed afterwards)"""
        self._chk(self.L.pomgpu_set_transport(self.h, None, _lib.TRANSPORT_FN(), None), "clear_transport")

    def rccl_unique_id(self, librccl: str | None = None) -> bytes:
        buf = ctypes.create_string_buffer(128)
        rc = self.L.pomgpu_rccl_unique_id(buf, librccl.encode() if librccl else None)
        if rc != 0:
            raise PomGpuError(f"pomgpu_rccl_unique_id failed with status {rc}")
        return buf.raw

    def exchange_rounds(self) -> int:
        return int(self.L.pomgpu_exchange_rounds(self.h))

    def exchange_rounds_side(self) -> int:
        """message rounds the library served on its second stream (beside kernels of the main stream)"""
        return int(self.L.pomgpu_exchange_rounds_side(self.h))

    def set_wide_external(self, on: bool, min_im: int, min_jm: int) -> bool:
        """collective; False when the tiles are too narrow (the per-point exchanges stay in use)"""
        rc = self.L.pomgpu_set_wide_external(self.h, 1 if on else 0, int(min_im), int(min_jm))
        if rc == -1 and on and self.L.pomgpu_last_error(self.h).decode().startswith("wide external mode: tiles"):
            return False
        self._chk(rc, "set_wide_external")
        return bool(on)

    def domain_stats(self, sums_only=False):
        """(vtot, atot, mtot, stot, tavg, savg, eavg, ekin) of advance.f:644-756, reduced on the device"""
        out = (ctypes.c_double * 8)()
        self._chk(self.L.pomgpu_domain_stats(self.h, out, 1 if sums_only else 0), "domain_stats")
        return tuple(out)

    def write_file(self, kind, path, title="", time_start="", im_global=None, jm_global=None, create=True, stats=None):
        """kind = "output" | "restart": the reference's NetCDF (CDF-2) files without PnetCDF (io_pnetcdf.F:57-410,
        :1661-2083); this tile's patch at (i_off+1, j_off+1) of the global grid"""
        st = self.st
        m = _lib.FileMeta(title.encode(), time_start.encode(), im_global or st.im, jm_global or st.jm, st.i_off + 1, st.j_off + 1,
                          1 if create else 0, (ctypes.c_double * 8)(*stats) if stats is not None else None)
        fn = self.L.pomgpu_write_output if kind == "output" else self.L.pomgpu_write_restart
        self._chk(fn(self.h, str(path).encode(), ctypes.byref(m)), "write_" + kind)

    def io_wait(self):
        """join the host thread that is writing the last output / restart file (sync, the next write and close do it too)"""
        self._chk(self.L.pomgpu_io_wait(self.h), "io_wait")

    def set_order_exchange(self, fn):
        """fn(send_east, n_east, send_north, n_north, recv_west, recv_south): device addresses (baropg_mcc's
        order2d_mpi / order3d_mpi, packed by the library) -- see extpom_amd.halo.Halo.device_order_hook"""
        def cb(user, se, ne, sn, nn, rw, rs):
            fn(se, ne, sn, nn, rw, rs)
        self._order_cb = _lib.ORDER_FN(cb)
        self._chk(self.L.pomgpu_set_order_exchange(self.h, self._order_cb, None), "set_order_exchange")

    # ---- hot path, reference names -------------------------------------------------------
    def _a(self, name):
        return self._p(self.st.field(name))

    def call(self, name, *fields_or_ints):
        fn = getattr(self.L, "pomgpu_" + name)
        args = [self._a(a) if isinstance(a, str) else a for a in fields_or_ints]
        self._chk(fn(self.h, *args), name)

    def advance(self):
        self.call("advance")

    def run(self, nsteps: int):
        self._chk(self.L.pomgpu_run(self.h, int(nsteps)), "run")

    def check_velocity(self):
        v = ctypes.c_double()
        i = ctypes.c_int()
        j = ctypes.c_int()
        self._chk(self.L.pomgpu_check_velocity(self.h, ctypes.byref(v), ctypes.byref(i), ctypes.byref(j)), "check_velocity")
        return v.value, i.value, j.value

    # ---- measurement ---------------------------------------------------------------------
    def prof_begin(self, only: str | None = None):
        self._chk(self.L.pomgpu_prof_filter(self.h, (only or "").encode()), "prof_filter")
        self._chk(self.L.pomgpu_prof_begin(self.h), "prof_begin")

    def prof_end(self) -> dict:
        self._chk(self.L.pomgpu_prof_end(self.h), "prof_end")
        out = {}
        for k in range(self.L.pomgpu_prof_count(self.h)):
            name = ctypes.c_char_p()
            n = ctypes.c_long()
            ms = ctypes.c_double()
            self.L.pomgpu_prof_get(self.h, k, ctypes.byref(name), ctypes.byref(n), ctypes.byref(ms))
            out[name.value.decode()] = (n.value, ms.value)
        return out


def gpu_finish_initial(st: PomState, **kw) -> PomState:
    """finish_initial() with the HIP dens / baropg (what the reference's initialize does with its own)."""
    from .cases import finish_initial
    g = PomGpu(st, **kw)

    def dens(s, si, ti, rho):
        g.upload(s)
        g.call("dens", si, ti, rho)
        g.download(s)

    def baropg(s):
        g.upload(s)
        g.call("baropg_mcc" if int(s.npg) == 2 else "baropg")
        g.download(s)

    finish_initial(st, dens, baropg)
    g.close()
    return st
