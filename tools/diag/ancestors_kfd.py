#!/usr/bin/env python3
"""developer diagnostic: which ancestors of this process hold the GPU device open?  (run as a rank under torch.distributed.run, or from
bench.py's pass children: POM_BENCH_REHEARSE=1 python bench.py --gpus 2 ... starts launcher -> elastic agent -> supervisor -> pass child)"""
import os
def fds(pid):
    out = set()
    try:
        for f in os.listdir(f"/proc/{pid}/fd"):
            try:
                t = os.readlink(f"/proc/{pid}/fd/{f}")
            except OSError:
                continue
            if "kfd" in t or "/dri/" in t:
                out.add(t)
    except OSError as e:
        return f"? ({e})"
    return sorted(out)
pid = os.getpid()
while pid > 1:
    try:
        cmd = open(f"/proc/{pid}/cmdline").read().replace("\0", " ")[:110]
        ppid = int(open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[1])
    except OSError:
        break
    print(f"[rank {os.environ.get('RANK', '-')}] pid {pid}: {fds(pid)}  {cmd}", flush=True)
    pid = ppid
