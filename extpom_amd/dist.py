"""One process per GPU: rank -> tile, process-group set-up (RCCL over xGMI = backend "nccl")."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .decomp import choose_tile_grid, local_size, make_tile


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment; single process if absent."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def cpu_barrier(group=None):
    """A barrier for processes that must not touch the GPU (rank supervisors, the CPU-baseline workers of bench.py):
    torch.distributed.barrier() on a gloo group initialises the GPU runtime in this torch build (measured on the GPU box,
    tools/diag/kfd_open.py: /dev/kfd opens at the first barrier, whatever *_VISIBLE_DEVICES says) -- a reduction of a CPU
    tensor does not."""
    t = torch.zeros(1)
    dist.all_reduce(t, group=group)


def tile_for_rank(rank: int, world: int, im_global: int, jm_global: int):
    nx, ny = choose_tile_grid(world, im_global, jm_global)
    if os.environ.get("POM_TILE_GRID"):          # developer switch: "2x4" forces nproc_x x nproc_y
        nx, ny = (int(v) for v in os.environ["POM_TILE_GRID"].lower().split("x"))
        assert nx * ny == world
    iml, jml = local_size(im_global, jm_global, nx, ny)
    return make_tile(rank, im_global, jm_global, iml, jml, n_proc=world)
