#!/usr/bin/env python3
"""developer diagnostic: at which point does a CPU-only torch.distributed (gloo) worker open the GPU device (/dev/kfd, /dev/dri/render*)?
The GPU boxes allow only a few processes with the device open; bench.py's all-cores CPU leg starts one worker per core."""
import os, sys
def fds(tag):
    n = []
    for f in os.listdir("/proc/self/fd"):
        try:
            t = os.readlink(f"/proc/self/fd/{f}")
        except OSError:
            continue
        if "kfd" in t or "/dri/" in t:
            n.append(t)
    print(f"[{os.environ.get('RANK','-')}] {tag}: {sorted(set(n))}", flush=True)
fds("start")
import numpy  # noqa
fds("numpy")
import torch
fds("import torch")
import torch.distributed as dist
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    be = os.environ.get("DIAG_BACKEND", "gloo")
    rank = int(os.environ["RANK"])
    dist.init_process_group(be, rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    fds(f"init {be}")
    t = torch.tensor([1.0]); dist.broadcast(t, 0)
    fds("broadcast")
    dist.all_reduce(t)
    fds("all_reduce")
    r = torch.zeros(4); s = torch.ones(4)
    for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, s, 1 - rank), dist.P2POp(dist.irecv, r, 1 - rank)]):
        q.wait()
    fds("batch_isend_irecv")
    box = [rank]; dist.broadcast_object_list(box, src=0)
    fds("broadcast_object_list")
    out = [None, None]; dist.all_gather_object(out, rank)
    fds("all_gather_object")
    store = dist.distributed_c10d._get_default_store(); store.add("k", 1)
    fds("store.add")
    if os.environ.get("DIAG_BARRIER", "1") == "1":
        dist.barrier()
        fds("barrier")
    dist.destroy_process_group()
    fds("destroy")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from extpom_amd.halo import Halo  # noqa
from oracle.pyoracle import OracleTile  # noqa
fds("halo + oracle imports")
