"""Halo exchange between tiles: the semantics of the reference's exchange2d_mpi / exchange3d_mpi
(reference pom/parallel_mpi.f:154-351) over torch.distributed point-to-point -- RCCL send/recv over
xGMI with backend "nccl" on MI355X, gloo on CPU (tests).

Ghost-cell contract (parallel_mpi.f:170-236): 1-cell rim; column im-1 goes to the eastern
neighbour's column 1 and column 2 to the western neighbour's column im; THEN row jm-1 goes to the
northern neighbour's row 1 and row 2 to the southern neighbour's row jm.  The north/south rows
include the ghost columns received in the first phase, which is how corners become correct.

Unlike the reference (one blocking send+recv pair per array per direction) all arrays exchanged
at one program point travel in ONE message per neighbour and phase, and both directions of a
phase are posted together (batch_isend_irecv = one grouped ncclSend/ncclRecv launch): the step
is latency-bound (~370 exchange points per internal step), not bandwidth-bound.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist


class _DevPtr:
    """Exposes a raw device address to torch through the CUDA array interface (ROCm honours it)."""

    def __init__(self, ptr: int, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class Halo:
    def __init__(self, tile, group=None, staged=False):
        """staged=True moves the packed edges through host memory (device tensors over a gloo group:
        used by the 2-ranks-on-one-GPU test; RCCL needs one GPU per rank)."""
        self.t = tile
        self.group = group
        self.staged = staged
        self.count = 0

    # ---- tensors --------------------------------------------------------------------------
    def wrap_device(self, ptr: int, nz: int, device) -> torch.Tensor:
        t = self.t
        return torch.as_tensor(_DevPtr(ptr, (nz, t.jm_local, t.im_local)), device=device)

    # ---- the exchange ---------------------------------------------------------------------
    def _phase(self, arrays, lo_nb, hi_nb, take, put):
        """lo_nb/hi_nb: neighbour ranks at the low/high end of this direction (-1 = none)."""
        ops, recv = [], {}
        bufs = []
        for nb, side in ((hi_nb, "hi"), (lo_nb, "lo")):
            if nb < 0:
                continue
            send = torch.cat([take(a, side).reshape(-1) for a in arrays]).contiguous()
            if self.staged:
                send = send.cpu()
            r = torch.empty_like(send)
            bufs.append(send)
            recv[side] = r
            ops.append(dist.P2POp(dist.isend, send, nb, group=self.group))
            ops.append(dist.P2POp(dist.irecv, r, nb, group=self.group))
        if not ops:
            return
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if self.staged:
            recv = {side: r.to(arrays[0].device) for side, r in recv.items()}
        for side, r in recv.items():
            off = 0
            for a in arrays:
                n = take(a, side).numel()
                put(a, side, r[off:off + n])
                off += n

    def exchange(self, arrays):
        """arrays: tensors shaped (nz, jm_local, im_local) (views of the levels to exchange)."""
        t = self.t
        im, jm = t.im, t.jm
        self.count += 1

        def take_x(a, side):   # east-going data is column im-1, west-going column 2 (1-based)
            return a[:, :jm, im - 2] if side == "hi" else a[:, :jm, 1]

        def put_x(a, side, v):  # from the east into column im, from the west into column 1
            if side == "hi":
                a[:, :jm, im - 1] = v.view(a.shape[0], jm)
            else:
                a[:, :jm, 0] = v.view(a.shape[0], jm)

        def take_y(a, side):
            return a[:, jm - 2, :im] if side == "hi" else a[:, 1, :im]

        def put_y(a, side, v):
            if side == "hi":
                a[:, jm - 1, :im] = v.view(a.shape[0], im)
            else:
                a[:, 0, :im] = v.view(a.shape[0], im)

        self._phase(arrays, t.n_west, t.n_east, take_x, put_x)
        self._phase(arrays, t.n_south, t.n_north, take_y, put_y)

    def order(self, send_e, send_n, recv_w, recv_s):
        """order2d_mpi / order3d_mpi (parallel_mpi.f:353-480): one-way, eastward and northward.
        send_e / send_n go to the eastern / northern neighbour, recv_w / recv_s come from the western /
        southern one (flat tensors; a missing neighbour skips that transfer)."""
        t = self.t
        ops, staged = [], []
        for buf, nb, fn in ((send_e, t.n_east, dist.isend), (recv_w, t.n_west, dist.irecv),
                            (send_n, t.n_north, dist.isend), (recv_s, t.n_south, dist.irecv)):
            if nb < 0:
                continue
            x = buf
            if self.staged:
                x = buf.cpu() if fn is dist.isend else torch.empty(buf.shape, dtype=buf.dtype)
                if fn is dist.irecv:
                    staged.append((x, buf))
            ops.append(dist.P2POp(fn, x, nb, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for host, dev in staged:
            dev.copy_(host)

    def numpy_order_hook(self):
        """for OracleTile(order=...): a (nz,ny,nx), ghost_w (nz,ny), ghost_s (nz,nx) numpy views"""
        def fn(a, ghost_w, ghost_s):
            nz, ny, nx = a.shape
            ta = torch.from_numpy(a)
            self.order(ta[:, :, nx - 3].contiguous().reshape(-1), ta[:, ny - 3, :].contiguous().reshape(-1),
                       torch.from_numpy(ghost_w).reshape(-1), torch.from_numpy(ghost_s).reshape(-1))
        return fn

    def device_order_hook(self, device):
        """for PomGpu.set_order_exchange: raw device addresses of the packed buffers"""
        def fn(send_e, n_e, send_n, n_n, recv_w, recv_s):
            w = lambda p, n: torch.as_tensor(_DevPtr(p, (n,)), device=device)
            self.order(w(send_e, n_e), w(send_n, n_n), w(recv_w, n_e), w(recv_s, n_n))
        return fn

    # ---- hooks ----------------------------------------------------------------------------
    def gpu_hook(self, device):
        """callback for PomGpu.set_exchange: device addresses + level counts"""
        def fn(ptrs, nzs):
            self.exchange([self.wrap_device(p, nz, device) for p, nz in zip(ptrs, nzs)])
        return fn

    def numpy_hook2d(self):
        import numpy as np  # noqa: F401

        def fn(a):          # a: numpy (ny, nx) view owned by the caller
            self.exchange([torch.from_numpy(a)[None]])
        return fn

    def numpy_hook3d(self):
        def fn(a):          # a: numpy (nz, ny, nx)
            self.exchange([torch.from_numpy(a)])
        return fn


class DeviceHalo:
    """The exchange hook for the HIP path: `Halo`'s semantics with the packing done by the library.

    Per program point ONE round: one pack launch (pomgpu_halo_pack8: every array's two edge columns, two edge
    rows and four corner cells into one contiguous buffer per neighbour), ONE batch_isend_irecv with up to eight
    neighbours (a grouped ncclSend/ncclRecv over xGMI), one unpack launch -- all on the stream the kernels run
    on; no per-array torch ops, no host-device synchronisation (RCCL path).  The ghost cells end up exactly as
    after the reference's two phases (pomgpu.h); `two_phase=True` keeps the reference's E/W-then-N/S rounds.
    Buffers are allocated once.
    """

    DIRS = ("w", "e", "s", "n", "sw", "se", "nw", "ne")

    def __init__(self, gpu, tile, device, group=None, staged=False, two_phase=False):
        self.g, self.t, self.group, self.staged = gpu, tile, group, staged
        self.count = 0
        self._vp = ctypes.c_void_p
        L = gpu.L
        kb = gpu.st.kb
        if two_phase:
            self._init_two_phase(gpu, tile, device, group, staged)
            return
        nb = dict(w=tile.n_west, e=tile.n_east, s=tile.n_south, n=tile.n_north, sw=tile.n_sw, se=tile.n_se, nw=tile.n_nw, ne=tile.n_ne)
        length = dict(w=tile.jm, e=tile.jm, s=tile.im, n=tile.im, sw=1, se=1, nw=1, ne=1)
        mk = lambda d: torch.empty(8 * kb * length[d], dtype=torch.float64, device=device)
        self.send = {d: mk(d) for d in self.DIRS if nb[d] >= 0}
        self.recv = {d: mk(d) for d in self.DIRS if nb[d] >= 0}
        tab = lambda bufs: (ctypes.c_void_p * 8)(*[bufs[d].data_ptr() if d in bufs else None for d in self.DIRS])
        to_tab, from_tab = tab(self.send), tab(self.recv)

        def hook(user, ptrs, nz, count):
            self.count += 1
            if not self.send:
                return
            total = 0
            for a in range(count):
                total += nz[a]
            if L.pomgpu_halo_pack8(gpu.h, ptrs, nz, count, to_tab):
                raise RuntimeError("pomgpu_halo_pack8 failed")
            ops, staged_recv = [], []
            for d in self.DIRS:
                if d not in self.send:
                    continue
                m = total * length[d]
                snd, rcv = self.send[d][:m], self.recv[d][:m]
                if staged:
                    snd = snd.cpu()
                    host = torch.empty_like(snd)
                    staged_recv.append((host, rcv))
                    rcv = host
                ops.append(dist.P2POp(dist.isend, snd, nb[d], group=group))
                ops.append(dist.P2POp(dist.irecv, rcv, nb[d], group=group))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for host, dev in staged_recv:
                dev.copy_(host)
            if L.pomgpu_halo_unpack8(gpu.h, ptrs, nz, count, from_tab):
                raise RuntimeError("pomgpu_halo_unpack8 failed")

        from .lib import EXCHANGE_FN
        self._cb = EXCHANGE_FN(hook)
        self._tabs = (to_tab, from_tab)
        gpu._chk(L.pomgpu_set_exchange(gpu.h, self._cb, None), "set_exchange")

    def _init_two_phase(self, gpu, tile, device, group, staged):
        L = gpu.L
        n = 8 * gpu.st.kb * max(tile.im, tile.jm)
        mk = lambda: torch.empty(n, dtype=torch.float64, device=device)
        self.buf = {(d, w, side): mk() for d in (0, 1) for w in ("s", "r") for side in ("lo", "hi")}
        self._vp = ctypes.c_void_p
        L = gpu.L

        def hook(user, ptrs, nz, count):
            self.count += 1
            total = 0
            for a in range(count):
                total += nz[a]
            for d in (0, 1):
                lo_nb, hi_nb = (tile.n_west, tile.n_east) if d == 0 else (tile.n_south, tile.n_north)
                if lo_nb < 0 and hi_nb < 0:
                    continue
                m = total * (tile.jm if d == 0 else tile.im)
                b = self.buf
                ptr = lambda t, on: self._vp(t.data_ptr()) if on else None
                rc = L.pomgpu_halo_pack(gpu.h, ptrs, nz, count, d, ptr(b[d, "s", "lo"], lo_nb >= 0), ptr(b[d, "s", "hi"], hi_nb >= 0))
                if rc:
                    raise RuntimeError("pomgpu_halo_pack failed")
                ops, staged_recv = [], []
                for nb, side in ((hi_nb, "hi"), (lo_nb, "lo")):
                    if nb < 0:
                        continue
                    snd, rcv = b[d, "s", side][:m], b[d, "r", side][:m]
                    if staged:
                        snd = snd.cpu()
                        host = torch.empty_like(snd)
                        staged_recv.append((host, rcv))
                        rcv = host
                    ops.append(dist.P2POp(dist.isend, snd, nb, group=group))
                    ops.append(dist.P2POp(dist.irecv, rcv, nb, group=group))
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                for host, dev in staged_recv:
                    dev.copy_(host)
                rc = L.pomgpu_halo_unpack(gpu.h, ptrs, nz, count, d, ptr(b[d, "r", "lo"], lo_nb >= 0), ptr(b[d, "r", "hi"], hi_nb >= 0))
                if rc:
                    raise RuntimeError("pomgpu_halo_unpack failed")

        from .lib import EXCHANGE_FN
        self._cb = EXCHANGE_FN(hook)
        gpu._chk(L.pomgpu_set_exchange(gpu.h, self._cb, None), "set_exchange")


class StagedMover:
    """A mover for pomgpu_set_transport on hosts where RCCL cannot connect the ranks (tests: several ranks on ONE
    GPU): the staging buffers travel through host memory over a gloo group.  Synchronises the kernels' stream on
    every round -- test infrastructure, not the production path (that is pomgpu_rccl_init)."""

    def __init__(self, gpu, tile, device, group=None):
        self.g, self.nb, self.device, self.group = gpu, gpu.neighbours8(tile), device, group

    def __call__(self, send, scount, recv, rcount):
        from .lib import OPP  # noqa: F401
        w = lambda p, n: torch.as_tensor(_DevPtr(p, (n,)), device=self.device)
        # the stream of THIS round (pomgpu_current_stream: the kernels' stream, or the library's second one, beside which the kernels' stream keeps running)
        cs = torch.cuda.ExternalStream(self.g.current_stream(), device=self.device)
        with torch.cuda.stream(cs):
            cs.synchronize()
            ops, staged = [], []
            for d in range(8):
                if self.nb[d] >= 0 and scount[d]:
                    ops.append(dist.P2POp(dist.isend, w(send[d], scount[d]).cpu(), self.nb[d], group=self.group, tag=d))
            for d in range(8):
                if self.nb[d] >= 0 and rcount[d]:
                    host = torch.empty(rcount[d], dtype=torch.float64)
                    staged.append((host, w(recv[d], rcount[d])))
                    ops.append(dist.P2POp(dist.irecv, host, self.nb[d], group=self.group, tag=OPP[d]))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for host, dev in staged:
                dev.copy_(host)
            cs.synchronize()


def dist_allmin(group=None):
    """the host's reduction behind pomgpu_transport_side_agree for callback movers whose ranks are torch.distributed
    processes: the minimum of every rank's own answer (one all-gather; any backend)"""
    def agree(mine: int) -> int:
        world = dist.get_world_size(group)
        box = [None] * world
        dist.all_gather_object(box, int(mine), group=group)
        return min(box)
    return agree


def rccl_library_path():
    """the librccl torch has already mapped (one copy per process), else the system one"""
    import os
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


def connect_rccl(gpu, tile, rank, world, group=None) -> bool:
    """pomgpu_rccl_init on every rank: rank 0 draws the unique id, torch.distributed (any backend) carries it.
    Two collective steps.  (1) Every rank checks that it can open librccl and find the entry points
    (pomgpu_rccl_available: no GPU call, nothing collective) and the answers are all-gathered: if any rank cannot, ALL
    return False before anyone has entered ncclCommInitRank -- a rank that failed there would leave the others waiting
    inside the collective for ever.  (2) ncclCommInitRank itself; a failure in it cannot be undone rank by rank, so it
    raises (the caller's deadline, bench.py, ends the job if a partner never arrives)."""
    lib = rccl_library_path()
    can = gpu.L.pomgpu_rccl_available(lib.encode() if lib else None) == 0
    cans = [None] * world
    dist.all_gather_object(cans, bool(can), group=group)
    if not all(cans):
        if rank == 0:
            print(f"connect_rccl: librccl unusable on ranks {[r for r, c in enumerate(cans) if not c]}", flush=True)
        return False
    box = [gpu.rccl_unique_id(lib) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    gpu.rccl_init(tile, box[0], rank, world, lib)
    return True
