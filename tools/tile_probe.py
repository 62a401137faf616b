#!/usr/bin/env python3
"""Developer measurement (not the bench contract): ONE tile of an N-tile decomposition of a bench workload on one
GPU, through the multi-tile code path (split kernels, library exchange, wide-halo external mode) with a mover that
copies the tile's own staging buffers back to it (a periodic stand-in for the neighbours: the arithmetic is not
the real run's, the kernels, launches and message rounds are).  Prints per-kernel device time and the wall time per
step, i.e. what a rank of the N-GPU run spends outside the xGMI transfers.

    python tools/tile_probe.py --tiles 8 --rank 5 [--workload basin2048] [--steps 5] [--no-wide]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--rank", type=int, default=5)
    ap.add_argument("--workload", default="basin2048")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--no-wide", action="store_true")
    ap.add_argument("--tune", action="store_true", help="pomgpu_tune_placement before the measurement")
    ap.add_argument("--ab", default="", help="NAME=V[,NAME=V...]: developer switches set on the LIVE context for every other block of steps -- "
                                             "the step with and without them, interleaved in one context (placement moves a kernel more than most changes do)")
    ap.add_argument("--ab-rounds", type=int, default=6)
    ap.add_argument("--rccl-self", action="store_true", help="the REAL transport instead of the stand-in mover: an RCCL communicator of one rank whose tile is its own "
                                                             "neighbour (periodic), so every round is a grouped ncclSend / ncclRecv on the kernels' or the second stream "
                                                             "-- RCCL's own launch path and kernels, without a link")
    ap.add_argument("--round-us", type=float, default=0.0, help="a MODEL of what a message round costs between GPUs: every round holds its stream for this "
                                                                "many microseconds (a spinning kernel) before its copies -- rounds between kernels then cost it on the step, "
                                                                "rounds on the second stream only if nothing runs beside them")
    a = ap.parse_args()
    import torch
    import bench
    from extpom_amd import dist as pdist
    from extpom_amd.halo import _DevPtr
    from extpom_amd.lib import OPP
    from extpom_amd.model import PomGpu
    case, im, jm, kb, desc = bench.WORKLOADS[a.workload]
    tile = pdist.tile_for_rank(a.rank, a.tiles, im, jm)
    if a.rccl_self:                                     # every neighbour the tile has is the one rank itself
        import dataclasses
        # (a direction without its opposite -- an edge tile, a 2-wide grid -- pairs with itself: messages between one pair of ranks match in the order
        # they are posted, and what leaves towards d has the size of what arrives from d)
        me = lambda n: 0 if n >= 0 else -1
        tile = dataclasses.replace(tile, rank=0, n_west=me(tile.n_west), n_east=me(tile.n_east), n_south=me(tile.n_south), n_north=me(tile.n_north),
                                   n_sw=me(tile.n_sw), n_se=me(tile.n_se), n_nw=me(tile.n_nw), n_ne=me(tile.n_ne))
    st = bench.build_state(a.workload, tile)
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    g = bench.gpu_initialise(st, 0, ts.cuda_stream)
    dev = torch.device("cuda", 0)
    nb = PomGpu.neighbours8(tile)
    w = lambda p, n: torch.as_tensor(_DevPtr(p, (n,)), device=dev)

    spin = 0
    if a.round_us > 0:                                  # calibrate torch's spinning kernel: cycles per microsecond on this box
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000000); torch.cuda.synchronize()
        e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
        spin = int(a.round_us * 10000000 / (e0.elapsed_time(e1) * 1e3))

    def mover(send, scount, recv, rcount):
        with torch.cuda.stream(torch.cuda.ExternalStream(g.current_stream())):   # the stream of the round (pomgpu.h): the kernels' or the second one
            if spin:
                torch.cuda._sleep(spin)
            for d in range(8):
                o = OPP[d]
                if nb[d] >= 0 and rcount[d]:
                    if nb[o] >= 0 and scount[o] == rcount[d]:
                        w(recv[d], rcount[d]).copy_(w(send[o], scount[o]))
                    elif scount[d] == rcount[d]:
                        w(recv[d], rcount[d]).copy_(w(send[d], scount[d]))
                    else:                               # one-way message without a matching buffer of the tile's own
                        w(recv[d], rcount[d]).zero_()

    if a.rccl_self:
        from extpom_amd.halo import rccl_library_path
        lib = rccl_library_path()
        g.rccl_init(tile, g.rccl_unique_id(lib), 0, 1, lib)
    else:
        g.set_transport(tile, mover, agree=lambda mine: mine, stream_ordered=True)    # a stand-in for N identical ranks: this rank's answer is everybody's; the mover enqueues on the round's stream, nothing waits for the device (as with RCCL)
    wide = False
    if not a.no_wide:
        tiles = [pdist.tile_for_rank(r, a.tiles, im, jm) for r in range(a.tiles)]
        wide = g.set_wide_external(True, min(t.im for t in tiles), min(t.jm for t in tiles))
    g.run(2)
    tuned = g.tune_placement(3, 10) if a.tune else None
    g.sync()
    g.prof_begin()
    g.run(1)
    prof = g.prof_end()
    r0, r0s = g.exchange_rounds(), g.exchange_rounds_side()
    t0 = time.perf_counter()
    g.run(a.steps)
    g.sync()
    dt = (time.perf_counter() - t0) / a.steps
    rounds = (g.exchange_rounds() - r0) / a.steps
    rounds_side = (g.exchange_rounds_side() - r0s) / a.steps
    ab = None
    if a.ab:
        sw = dict(kv.split("=", 1) for kv in a.ab.split(","))
        acc = {"default": [], a.ab: []}
        for _ in range(a.ab_rounds):
            for tag in acc:
                for k, v in sw.items():
                    g.switch(k, v if tag != "default" else None)
                g.run(1); g.sync()
                t1 = time.perf_counter()
                g.run(a.steps)
                g.sync()
                acc[tag].append((time.perf_counter() - t1) / a.steps * 1e3)
        for k in sw:
            g.switch(k, None)
        ab = {tag: {"min": round(min(v), 3), "median": round(sorted(v)[len(v) // 2], 3)} for tag, v in acc.items()}
    prof.pop("phase_step", None); prof.pop("phase_external", None)     # brackets around other brackets (pomgpu.h): not kernels
    msg = prof.pop("msg_round", (0, 0.0))
    msg_side = prof.pop("msg_round_side", (0, 0.0))
    share = sorted(((k, v[0], v[1]) for k, v in prof.items()), key=lambda kv: -kv[2])
    print(json.dumps({"workload": desc, "workload_key": a.workload, "library_build_id": g.L.pomgpu_build_id().decode(), "tiles": f"{tile.nproc_x}x{tile.nproc_y}", "rank": a.rank, "tile": f"{tile.im}x{tile.jm}x{kb}",
                      "wide": bool(wide), "placement": tuned, "ab_ms_per_step_wall": ab, "ms_per_step_wall": round(dt * 1e3, 3), "message_rounds_per_step": rounds,
                      "kernel_ms_sum": round(sum(v[2] for v in share), 3), "stand_in_mover_ms": round(msg[1], 3),
                      "message_rounds_on_side_stream_per_step": rounds_side, "modelled_round_latency_us": a.round_us, "transport": "RCCL, one rank that is its own neighbour" if a.rccl_self else "stand-in mover (device copies on the round's stream)", "stand_in_mover_side_ms": round(msg_side[1], 3),
                      "kernels": {k: [n, round(ms, 3)] for k, n, ms in share[:45]}}))
    g.close()


if __name__ == "__main__":
    main()
