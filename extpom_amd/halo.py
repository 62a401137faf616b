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
