#!/usr/bin/env python3
"""Benchmark of the hot path: whole internal steps of the mode-split time step (one `advance`:
lateral_viscosity, mode_interaction, isplit external substeps, mode_internal, check_velocity).

    python bench.py --gpus N --steps K --warmup W [--workload basin2048|seamount256|basin1024|...]

N=1 runs in this process.  N>1: one rank per GPU, the global grid split into N tiles (strong scaling: the
global grid is fixed) -- either started by the caller under torch.distributed.run (WORLD_SIZE set), or, when
`python bench.py --gpus N` is called plainly, by this script itself: the parent starts
`python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process before it
touches the GPU, relays the child's output and exits with its code (reference launch shape: pom.sh:1
`mpiexec -n 8`, parallel_mpi.f:6-20).  Rank 0 prints ONE JSON line.  `value` = im_global*jm_global*kb*K / max-over-ranks wall time of the K timed
steps, state resident in HBM before the timed region starts.

Extra objects on the same line:
  roofline      dominant kernel (largest share of device time in a profiled step): ALGORITHMIC bytes
                per launch / its mean launch duration measured with HIP events on the kernels' own
                stream inside the timed region (only that kernel is bracketed there).
  cpu_baseline  the CPU oracle (plain-C restatement, bit-identical to the reference build) timed on
                this box's host, 1 core, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (case, im, jm, kb, description)
    "basin2048": ("basin", 2048, 1536, 50, "closed basin 2048x1536x50 fp64 (BASELINE configs[3] grid)"),
    "basin1024": ("basin", 1024, 1024, 40, "closed basin 1024x1024x40 fp64 (BASELINE configs[2] grid)"),
    "seamount256": ("seamount", 256, 256, 30, "seamount 256x256x30 fp64 (BASELINE configs[1])"),
    "seamount65": ("seamount", 65, 49, 21, "seamount 65x49x21 fp64 (BASELINE configs[0])"),
}
NML = dict(dte=6.0, isplit=30, mode=3, nadv=2, nitera=1, npg=1)

# ALGORITHMIC traffic per kernel launch in full 3-D array passes (distinct arrays a launch must read
# + write once; 2-D arrays cost 1/kb and are not counted) -- DESIGN.md "Kernels" derives each row from
# SURVEY 8(a)/(d).  bytes per launch = passes * 8 B * im*jm*kb of the tile.
KERNEL_PASSES = {
    "k_profq": 23,           # a10 + a11, one tile (production term and the q2/q2l Asselin filter inside): R kq,km,kh,t,s,rho,q2b,q2lb,q2,q2l,uf,vf,u,v W q2b,q2lb,q2,q2l,l,dtef,kq,km,kh
    "k_profq/tiles": 22,     # several tiles: prod comes from k_profq_prod (exchanged): R ...,prod instead of u,v
    "k_advct_col": 7,        # a2, one tile: R u,v,ub,vb,aam W advx,advy
    "k_advq2_col": 10,       # a9 for q2 and q2l together: R q2,q2b,q2l,q2lb,u,v,w,aam W uf,vf
    "k_advt2x2_col": 10,     # a13 for T and S together: R tb,tclim,sb,sclim,u,v,w,aam W uf,vf
    "k_advq_col": 7,         # a9, one tile: R q,qb,u,v,w,aam W qf
    "k_advt2_col": 7,        # a13: R fb,fclim,u,v,w,aam W ff
    "k_ts_update": 17,       # a15+restore_interior+a16: R uf,vf,t,tb,s,sb,tclim,sclim,4 restore fields (taurstrb/f are known scalars once the library has loaded a record) W t,tb,s,sb,rho
    "k_profq_prod": 9,       # R km,kh,t,s,rho,u,v (+1 k-shifted reuse counted once) W prod  -> 7R+1W (+1)
    "k_advq_flux": 7,        # a9 first half: R q,qb,u,v,aam W xflux,yflux
    "k_advq_step": 6,        # a9 second half: R q,qb,w,xflux,yflux W qf
    "k_advct_a": 8,          # a2: R u,v,ub,vb,aam W curv,xflux,yflux
    "k_advct_b": 11,         # R curv,xflux,yflux,u,v,ub,vb,aam W advx,xflux',yflux'
    "k_advct_c": 6,          # R curv,xflux',yflux',u (+v) W advy
    "k_advu_profu": 9,       # a17+a18: R w,u,v,advx,drhox,ub,vb(kbm1),km W uf
    "k_advv_profv": 9,
    "k_uv_filter": 10,       # a19: R uf,ub,u,vf,vb,v W ub,u,vb,v
    "k_proft": 3,            # a14: R f,kh W f
    "k_proft_reg": 3,
    "k_profu_reg": 3,        # a18: R uf,km W uf
    "k_profv_reg": 3,
    "k_advuv_col": 11,       # a17: R w,u,v,ub,vb,advx,advy,drhox,drhoy W uf,vf
    "k_aam_pair": 3,
    "k_realvertvl_col": 4,
    "k_q_filter": 10,
    "k_restore": 13,         # R trstrb/f,srstrb/f,taurstrb/f,t,tb,s,sb W trstr,srstr,taurstr,t,tb,s,sb (not in SURVEY's 133)
    "k_dens": 3,             # a16
    "k_baropg": 4,           # a3
    "k_roundtrip": 3,
    "k_aam": 3,              # a1
    "k_vint": 5,             # a4
    "k_int_uvmean": 4,       # a7
    "k_vertvl": 3,           # a8
    "k_realvertvl": 4,       # a20
}
P_STEP = 133                 # algorithmic 3-D passes per internal step (mode=3 nadv=2 nitera=1)
HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0        # measured float4-copy ceiling (same guide); SURVEY 8(d): report both


def _libc_version():
    import ctypes
    try:
        f = ctypes.CDLL("libc.so.6").gnu_get_libc_version
        f.restype = ctypes.c_char_p
        return "glibc " + f().decode()
    except Exception:                     # noqa: BLE001
        return "unknown"


def build_state(workload, tile):
    from extpom_amd.cases import make_case
    case, im, jm, kb, _ = WORKLOADS[workload]
    return make_case(case, im, jm, kb, tile=tile, **NML)


def gpu_initialise(st, device, stream, libpath=None):
    """the reference's initialisation tail (dens, baropg, bottom friction ...) with the HIP kernels"""
    from extpom_amd.cases import finish_initial
    from extpom_amd.layout import P3
    from extpom_amd.model import PomGpu
    g = PomGpu(st, device=device, stream=stream, libpath=libpath)
    P = lambda a: __import__("ctypes").c_void_p(a.ctypes.data)

    def dens(s, si, ti, rho):
        g.call("dens", si, ti, rho)
        g._chk(g.L.pomgpu_download_3d(g.h, P3[rho], P(s.field(rho))), "download")

    def baropg(s):
        g.call("baropg_mcc" if int(s.npg) == 2 else "baropg")
        for f in ("drhox", "drhoy", "rho"):
            g._chk(g.L.pomgpu_download_3d(g.h, P3[f], P(s.field(f))), "download")

    finish_initial(st, dens, baropg)
    g.upload(st)
    return g


def _sample_dims(workload):
    case, im, jm, kb, _ = WORKLOADS[workload]
    return case, max(65, im // 8), max(49, jm // 8), kb


def _cpu_sample(workload, seconds=12.0, reference=False):
    """ONE host core, bounded sample: same case/namelist/kb on a 1/8 x 1/8 horizontal grid.  reference=False: the
    plain-C oracle; True: the flang-built UNMODIFIED reference (oracle/_ref, built where /root/reference exists and
    shipped prebuilt), driven subroutine by subroutine in the order of its own `advance` (oracle/refharness.py)."""
    from extpom_amd.cases import make_case
    from oracle.pyoracle import OracleTile, oracle_finish_initial
    case, sim, sjm, kb = _sample_dims(workload)
    st = make_case(case, sim, sjm, kb, **NML)
    oracle_finish_initial(st)
    if reference:
        from oracle.refharness import RefLib
        lib = RefLib(sim, sjm, kb)
        lib.put(st)
        it = [0]

        def step():
            it[0] += 1
            lib.con["iint"][0] = it[0]
            lib.advance()
    else:
        ot = OracleTile(st)
        step = lambda: ot.run(1)
    step(); step()
    t0 = time.perf_counter()
    n = 0
    while n < 4 or (time.perf_counter() - t0 < seconds and n < 400):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return sim * sjm * kb * n / dt, f"{case} {sim}x{sjm}x{kb}", n, dt


def _reference_sample(workload):
    """child-process body of cpu_reference: the reference keeps its work arrays on the stack (automatic arrays: 11 x
    (im,jm,kb) doubles in profq alone, solver.f:1224-1230), so it runs in a thread with a stack to match"""
    import threading
    box = {}

    def work():
        box["r"] = _cpu_sample(workload, reference=True)
    threading.stack_size(1 << 30)
    th = threading.Thread(target=work)
    th.start()
    th.join()
    return box["r"]


def cpu_reference(workload):
    """the reference itself on one core, if its build for the sample's size travelled with the repo; in a child
    process, so that nothing it does can take the bench down"""
    import subprocess
    from oracle.refharness import have_ref
    case, sim, sjm, kb = _sample_dims(workload)
    if not have_ref(sim, sjm, kb):
        return None
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-sample", "--reference", "--workload", workload],
                       capture_output=True, text=True, timeout=240)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": d["value"], "unit": "cell-updates/s", "cores": 1, "kind": "reference",
            "sample": f"{d['what']}, {d['n']} internal steps of the unmodified reference (solver.f advance.f bounds_forcing.f, "
                      f"AMD flang -O2), {d['seconds']:.1f} s"}


def cpu_baseline(workload):
    v, what, n, dt = _cpu_sample(workload)
    return {"value": v, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"{what}, {n} internal steps of the plain-C oracle (gcc -O2, bit-identical to the flang-built reference), {dt:.1f} s"}


def cpu_baseline_all_cores(workload):
    """the same sample on every host core at once (one independent copy per core, as the reference's MPI ranks
    would each hold a tile): the host's aggregate rate, memory-bandwidth contention included.  The copies are plain
    child processes of this script (`--cpu-sample`), each bounded by a timeout."""
    import subprocess
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))        # a 1-GPU box's CPU share
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-sample", "--workload", workload],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
    res = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
            res.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:                                          # noqa: BLE001 -- a lost copy only lowers the sum
            p.kill()
    if not res:
        return None
    return {"value": sum(r["value"] for r in res), "unit": "cell-updates/s", "cores": len(res), "kind": "port",
            "sample": f"{len(res)} concurrent copies of {res[0]['what']} (one process per core), "
                      f"{min(r['n'] for r in res)}-{max(r['n'] for r in res)} internal steps each"}


def side_config(workload, device, stream, steps=40, warmup=5):
    """a second workload on the same GPU, timed like the main one (state resident, K steps between syncs)"""
    case, im, jm, kb, desc = WORKLOADS[workload]
    from extpom_amd import dist as pdist
    st = build_state(workload, pdist.tile_for_rank(0, 1, im, jm))
    g = gpu_initialise(st, device, stream)
    g.run(warmup)
    g.sync()
    t0 = time.perf_counter()
    g.run(steps)
    g.sync()
    dt = time.perf_counter() - t0
    g.get_con()
    err = int(st.error_status)
    g.close()
    return {"workload": desc, "ms_per_step": dt / steps * 1e3, "value": im * jm * kb * steps / dt, "unit": "cell-updates/s", "steps": steps,
            "step_algorithmic_GBps": round(P_STEP * 8.0 * im * jm * kb / (dt / steps) / 1e9, 1), "error_status": err}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1), pass its output through and return its exit code.  The parent never initialises
    the GPU, and nothing is re-executed in place."""
    import socket
    import subprocess
    if os.environ.get("POM_BENCH_REHEARSE") != "1":
        import torch
        have = torch.cuda.device_count()                       # counting devices does not initialise the GPU in this process
        if have < n:
            print(f"bench: --gpus {n} but {have} GPU(s) visible -- one rank per GPU, no measurement", file=sys.stderr)
            return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout is passed on line by line: the JSON line to stdout, anything a library wrote there (gloo's
    # "[Gloo] Rank ..." banners in a rehearsal) to stderr, so that stdout holds the one line the contract asks for
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:
        t = line.strip()
        is_json = t.startswith("{") and t.endswith("}")
        (sys.stdout if is_json else sys.stderr).write(line)
        (sys.stdout if is_json else sys.stderr).flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("POM_BENCH_WORKLOAD", "basin2048"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--side-config", action="store_true",
                    help="N=1: also time BASELINE configs[1] (seamount 256x256x30) in the same run; off by default so that a profile "
                         "of the default command holds the headline workload's kernels only")
    ap.add_argument("--reference", action="store_true", help="internal: with --cpu-sample, time oracle/_ref instead of the oracle")
    ap.add_argument("--cpu-sample", action="store_true", help="internal: run the one-core oracle sample and print it (no GPU)")
    ap.add_argument("--profile-all", action="store_true", help="bracket every kernel with events in the timed region")
    ap.add_argument("--storage", choices=["f64", "f32"], default="f64",
                    help="f32: BASELINE configs[4]'s STUDY variant (libpomgpu_f32.so: 3-D arrays stored as fp32, arithmetic and the external "
                         "mode fp64; one GPU).  Not a parity path (DESIGN.md section 8) and never the headline: the line says so in `dtype`")
    args = ap.parse_args()
    if args.cpu_sample:
        v, what, n, dt = _reference_sample(args.workload) if args.reference else _cpu_sample(args.workload)
        print(json.dumps({"value": v, "what": what, "n": n, "seconds": dt}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))                      # nothing in this process has touched the GPU

    import torch
    from extpom_amd import decomp, dist as pdist
    # POM_BENCH_REHEARSE=1: developer rehearsal of the N > 1 code path on a ONE-GPU box -- every rank on GPU 0,
    # gloo with host-staged halos instead of RCCL (RCCL refuses two ranks on one device).  Not a measurement.
    rehearse = os.environ.get("POM_BENCH_REHEARSE") == "1"
    rank, world, local = pdist.init("gloo" if rehearse else None)
    if rehearse:
        local = 0
    if world != args.gpus:
        if rank == 0:
            print(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench: no GPU visible -- the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local)
    case, im, jm, kb, desc = WORKLOADS[args.workload]
    tile = pdist.tile_for_rank(rank, world, im, jm)
    st = build_state(args.workload, tile)
    stream = None
    if world > 1:
        # kernels and the exchange's pack/unpack must share ONE stream (torch's default stream has
        # handle 0, which the C ABI reads as "create your own")
        ts = torch.cuda.Stream()
        torch.cuda.set_stream(ts)
        stream = ts.cuda_stream
    f32 = args.storage == "f32"
    if f32 and world > 1:
        print("bench: --storage f32 is a one-GPU study (the fp32-storage variant has no tile exchange)", file=sys.stderr)
        sys.exit(2)
    from extpom_amd import lib as _L
    g = gpu_initialise(st, local, stream, _L.LIBPATH_F32 if f32 else None)
    build_id = g.L.pomgpu_build_id().decode()
    exchange = "none"
    # N > 1: a rank that is lost, or a message round whose partner never posts, would leave the others waiting inside
    # RCCL for ever.  Every phase -- connecting, the first steps (RCCL sets its channels up lazily), the timed steps --
    # gets a deadline; a rank that misses one says where it was and leaves with a non-zero code, which ends the job.
    phase = ["start", None]

    def deadline(name, seconds):
        import threading
        if phase[1] is not None:
            phase[1].cancel()
            phase[1] = None
        phase[0] = name
        if world > 1 and seconds:
            def expired():
                print(f"bench[{rank}]: '{name}' did not complete within {seconds:.0f} s -- giving up", file=sys.stderr, flush=True)
                os._exit(4)
            phase[1] = threading.Timer(seconds, expired)
            phase[1].daemon = True
            phase[1].start()

    deadline("connect", 300.0)
    if world > 1:
        # The library serves every exchange point itself: pack -> one grouped ncclSend/ncclRecv round (RCCL over
        # xGMI, enqueued on the kernels' stream by the library, no Python in the loop) -> unpack; and the 2-D
        # external mode runs on a wide-halo copy of the tile (one exchange per internal step instead of ~180).
        from extpom_amd import halo as H
        dev = torch.device("cuda", local)
        if rehearse:
            g.set_transport(tile, H.StagedMover(g, tile, dev), agree=H.dist_allmin())
            exchange = "library exchange, host-staged mover (rehearsal)"
        else:
            if H.connect_rccl(g, tile, rank, world):
                exchange = "library exchange, native RCCL send/recv on the kernels' stream (+ a second stream and communicator for the early part of the wide exchange and wr)"
            elif os.environ.get("POM_BENCH_ALLOW_P2P") == "1":   # developer switch: torch.distributed's RCCL P2P carries the same messages
                print(f"bench[{rank}]: native RCCL transport unavailable; using torch.distributed P2P", file=sys.stderr)
                bench_halo = H.DeviceHalo(g, tile, dev)
                g.set_order_exchange(H.Halo(tile).device_order_hook(dev))   # npg = 2 only
                exchange = "torch.distributed batch_isend_irecv (RCCL) per exchange point"
            else:
                # a scaling number measured on a silent substitute would be worthless: no line, non-zero exit
                print(f"bench[{rank}]: the library's RCCL transport could not connect the {world} ranks -- no measurement",
                      file=sys.stderr, flush=True)
                sys.exit(5)
        if exchange.startswith("library") and os.environ.get("POM_BENCH_WIDE", "1") != "0":
            tiles = [pdist.tile_for_rank(r, world, im, jm) for r in range(world)]
            if g.set_wide_external(True, min(t.im for t in tiles), min(t.jm for t in tiles)):
                exchange += "; wide-halo external mode"

    def barrier():
        g.sync()
        if world > 1:
            torch.distributed.barrier()

    # warm-up, then a profiled step to find the dominant kernel.  At N > 1 the first rounds also set up the RCCL
    # connections
    deadline("warm-up steps", 300.0)
    g.run(max(args.warmup, 1))
    barrier()
    deadline("profiled step", 120.0)
    g.prof_begin()
    g.run(1)
    prof = g.prof_end()
    # dominant kernel = the most expensive 3-D kernel (the 2-D external-mode kernels have no 3-D pass count)
    cand = {k: v for k, v in prof.items() if k in KERNEL_PASSES} or prof
    dom = max(cand.items(), key=lambda kv: kv[1][1])[0] if cand else None
    barrier()

    # the timed region: exactly K steps, only the dominant kernel bracketed by events
    g.prof_begin(only=None if args.profile_all else dom)
    barrier()
    deadline("timed steps", 120.0 + 2.0 * args.steps)          # generous: a step takes tens of milliseconds
    rounds0, rounds0s = g.exchange_rounds(), g.exchange_rounds_side()
    t0 = time.perf_counter()
    g.run(args.steps)
    g.sync()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    timed = g.prof_end()
    deadline("wrap-up", 300.0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    g.get_con()
    err = int(st.error_status)

    if rank == 0:
        cells = im * jm * kb
        tile_cells = tile.im_local * tile.jm_local * kb
        ms = dt / args.steps * 1e3
        value = cells * args.steps / dt
        nl, tms = timed.get(dom, (0, 0.0))
        # on tiles served by the library's own exchange k_profq forms the production term itself, as on one tile
        split = world > 1 and not exchange.startswith("library") and dom + "/tiles" in KERNEL_PASSES
        passes = KERNEL_PASSES.get(dom + "/tiles" if split else dom)
        roof = None
        if nl and passes:
            ach = passes * (4.0 if f32 else 8.0) * tile_cells / (tms / nl * 1e-3) / 1e9
            # HBM-side bytes per launch from the PMC counters (profiles/traffic.json, tools/make_traffic_json.py): quoted only for
            # the library build the counters were taken from -- after a kernel change the stored number says nothing
            traffic, traffic_note = None, None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tf) and not f32:
                tj = json.load(open(tf))
                rec = tj.get(f"{args.workload}/{world}/{dom}")
                if rec and tj.get("_build_id") == build_id:
                    traffic = rec["bytes_per_launch"]
                elif rec:
                    traffic_note = f"profiles/traffic.json holds counters of build {tj.get('_build_id')}, this library is {build_id}: not quoted"
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "frac_of_measured_copy_ceiling": round(ach / HBM_COPY_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": passes * (4 if f32 else 8) * tile_cells, "launches": nl,
                    "avg_launch_ms": round(tms / nl, 4)}
            if traffic_note:
                roof["traffic_note"] = traffic_note
        # internal (3-D) mode alone, from the all-kernels profiled step: everything but the 2-D kernels
        ext = ("k_ext_", "k_advave_", "k_modeint_tail", "k_int_tail", "k_check_velocity", "k_check_areas", "k_copy2", "k_bcond1")
        msg_ms = prof.pop("msg_round", (0, 0.0))[1]           # the message rounds of the profiled step (N > 1): not a kernel
        msg_side_ms = prof.pop("msg_round_side", (0, 0.0))[1] # ... those on the library's second stream: beside kernels, not between them
        prof.pop("phase_step", None); prof.pop("phase_external", None)
        int_ms_one = sum(v[1] for k, v in prof.items() if not k.startswith(ext))    # the profiled step: sum of kernel durations
        ext_ms_one = sum(v[1] for k, v in prof.items() if k.startswith(ext))
        # the TIMED region: every step bracketed as a whole and around its external substeps (events on the kernels' stream):
        # K-step means, measured while only the dominant kernel carries events of its own
        nst, step_dev_ms = timed.get("phase_step", (0, 0.0))
        nex, ext_dev_ms = timed.get("phase_external", (0, 0.0))
        if nst == args.steps and nex == args.steps:
            ext_ms = ext_dev_ms / nex
            int_ms = step_dev_ms / nst - ext_ms
            int_note = (f"mean over the {args.steps} timed steps on rank 0: device time of the step (events around pomgpu_advance on the kernels' stream) minus "
                        "the device time of its isplit external substeps; advave / the 2-D tail of mode_interaction (once per step) count as internal here")
        else:
            int_ms, ext_ms = int_ms_one, ext_ms_one
            int_note = "sum of 3-D kernel durations of one profiled step on rank 0 (its tile only)"
        BPV = 4.0 if f32 else 8.0                              # bytes per stored 3-D value
        step_gbs = P_STEP * BPV * cells / (dt / args.steps) / 1e9
        share = sorted(((k, v[1]) for k, v in prof.items()), key=lambda kv: -kv[1])
        tot = sum(v for _, v in share) or 1.0
        out = {
            "metric": "3D cell-updates/sec (whole internal step incl. the isplit external substeps)",
            "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32-storage of the 3-D arrays / f64 arithmetic and external mode (STUDY variant, not a parity path)" if f32 else "f64",
            "data": "synthetic",
            "config": {"workload": desc + f", mode=3 nadv=2 nitera=1 npg=1 dte=6 isplit=30", "tiles": f"{tile.nproc_x}x{tile.nproc_y}",
                       "tile": f"{tile.im_local}x{tile.jm_local}x{kb}", "global_cells": cells, "exchange": exchange,
                       "message_rounds_per_step": (g.exchange_rounds() - rounds0) / args.steps if world > 1 else 0,
                       "message_rounds_per_step_on_side_stream": (g.exchange_rounds_side() - rounds0s) / args.steps if world > 1 else 0},
            "roofline": roof,
            "step_algorithmic_GBps": round(step_gbs, 1), "step_frac_of_peak": round(step_gbs / HBM_PEAK_GBS, 4),
            "step_frac_of_measured_copy_ceiling": round(step_gbs / HBM_COPY_GBS, 4),
            "internal_mode": {"device_ms_per_step": round(int_ms, 3), "cell_updates_per_s": (tile_cells / (int_ms * 1e-3)) if int_ms else None,
                              "algorithmic_GBps": round(P_STEP * BPV * tile_cells / (int_ms * 1e-3) / 1e9, 1) if int_ms else None,
                              "frac_of_peak": round(P_STEP * BPV * tile_cells / (int_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if int_ms else None,
                              "frac_of_measured_copy_ceiling": round(P_STEP * BPV * tile_cells / (int_ms * 1e-3) / 1e9 / HBM_COPY_GBS, 4) if int_ms else None,
                              "note": int_note, "profiled_step_kernel_sum_ms": round(int_ms_one, 3)},
            "external_mode": {"device_ms_per_step": round(ext_ms, 3), "profiled_step_kernel_sum_ms": round(ext_ms_one, 3)},
            "device_ms_per_step": round(step_dev_ms / nst, 3) if nst else None,
            "library_build_id": build_id,
            # all kernels of one profiled step on rank 0 (without the RCCL transfers): at N > 1, ms_per_step minus this is what
            # the message rounds and the waiting for neighbours cost
            "kernel_ms_sum_rank0": round(sum(v[1] for v in prof.values()), 3),
            "message_rounds_ms_rank0": round(msg_ms, 3),          # transfers + waiting for the neighbours, same profiled step: EXPOSED (kernels' stream)
            "message_rounds_side_stream_ms_rank0": round(msg_side_ms, 3),   # HIDDEN: on the second stream, beside lateral_viscosity / the next step
            "kernel_time_share": {k: round(v / tot, 3) for k, v in share[:8]},
            "kernel_ms_per_step": {k: round(v, 3) for k, v in share[:40]},
            "error_status": err,
            "host_libc": _libc_version(),     # bit-parity with the reference leans on this libm's pow (THIRD_PARTY_NOTICES.md)
        }
        if not args.no_cpu_baseline and world == 1:
            port = cpu_baseline(args.workload)
            ref = None
            try:
                ref = cpu_reference(args.workload)
            except Exception as e:                                 # noqa: BLE001 -- e.g. the MPI runtime the reference links is absent
                print(f"bench: reference CPU baseline unavailable ({e})", file=sys.stderr)
            out["cpu_baseline"] = ref or port                       # the reference itself where its build is present
            if ref:
                out["cpu_baseline_port"] = port
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.workload)
        if world == 1 and args.workload == "basin2048" and args.side_config:
            # BASELINE configs[1] (seamount 256x256x30) in the same run: a launch-latency-bound size, reported beside
            # the headline grid so that both ends of the size range are on the line
            g.close()
            g = None
            out["config1_seamount256"] = side_config("seamount256", local, stream)
        print(json.dumps(out))
    if g is not None:
        g.close()
    if world > 1:
        torch.distributed.barrier()
        deadline("done", 0)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
