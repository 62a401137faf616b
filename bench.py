#!/usr/bin/env python3
"""Benchmark of the hot path: whole internal steps of the mode-split time step (one `advance`:
lateral_viscosity, mode_interaction, isplit external substeps, mode_internal, check_velocity).

    python bench.py --gpus N --steps K --warmup W [--workload basin2048|seamount256|basin1024|...]

N=1 runs in this process.  N>1: one rank per GPU, the global grid split into N tiles (strong scaling: the
global grid is fixed) -- started by the caller under torch.distributed.run (WORLD_SIZE set), or, when
`python bench.py --gpus N` is called plainly, by this script itself (N child processes with the environment
torch.distributed.run would give them; reference launch shape: pom.sh:1 `mpiexec -n 8`, parallel_mpi.f:6-20).
Either way a rank's process is only a SUPERVISOR: it never touches the GPU.  It runs the measurement as a sequence of
PASSES, each a fresh child process per rank with a rendezvous of its own and deadlines of its own:
  1. the primary tile grid, every message round on the kernels' stream (one stream, one communicator: POMGPU_NO_OVERLAP=1);
  2. the primary tile grid with the second stream and the second (split) communicator -- north_star's overlapped halo exchange;
  3. the alternate tile grid (2x4 beside 1x8, parallel_mpi.f:54-65) under whichever of 1 / 2 was faster.
A pass that hangs or dies is a child that exits non-zero: its supervisors end it, say in which phase it stopped, and the
next pass starts from clean processes; the line's `value` is the fastest completed pass on the primary grid, `passes`
lists them all.  Rank 0 prints ONE JSON line.  `value` = im_global*jm_global*kb*K / max-over-ranks wall time of the K timed
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


def _tile_probe_bounds(workload):
    """profiles/tile_probe_bounds.json (tools/make_tile_bounds.py): ms per step of one tile of the 2 / 4 / 8-tile split and of the single GPU in the same job"""
    try:
        with open(os.path.join(ROOT, "profiles", "tile_probe_bounds.json")) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


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


def _cpu_run(workload, full, seconds=12.0, reference=False):
    """ONE host core.  full=True: the bench's OWN grid and state (identical inputs: same case, namelist, im x jm x kb), internal
    steps 1-2, of which step 2 is timed (step 1 skips the 3-D body, advance.f:362; one full-grid step is ~40 s of one core, and
    the default run has to stay within minutes); full=False: a bounded sample on a
    1/8 x 1/8 horizontal grid for `seconds`.  reference=False: the plain-C oracle; True: the flang build of the reference's own
    sources (oracle/_ref, present only where /root/reference was at build time), driven in the order of its own `advance`."""
    from extpom_amd.cases import make_case
    from oracle.pyoracle import OracleTile, oracle_finish_initial
    case, im, jm, kb, _ = WORKLOADS[workload]
    if not full:
        case, im, jm, kb = _sample_dims(workload)
    st = make_case(case, im, jm, kb, **NML)
    oracle_finish_initial(st)
    if reference:
        from oracle.refharness import RefLib
        lib = RefLib(im, jm, kb)
        lib.put(st)
        it = [0]

        def step():
            it[0] += 1
            lib.con["iint"][0] = it[0]
            lib.advance()
    else:
        ot = OracleTile(st)
        step = lambda: ot.run(1)
    step()
    if not full:
        step()
    t0 = time.perf_counter()
    n = 0
    while (n < 1) if full else (n < 4 or (time.perf_counter() - t0 < seconds and n < 400)):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return im * jm * kb * n / dt, f"{case} {im}x{jm}x{kb}", n, dt


def _reference_sample(workload):
    """child-process body of cpu_reference: the reference keeps its work arrays on the stack (automatic arrays: 11 x
    (im,jm,kb) doubles in profq alone, solver.f:1224-1230), so it runs in a thread with a stack to match"""
    import threading
    box = {}

    def work():
        box["r"] = _cpu_run(workload, False, reference=True)
    threading.stack_size(1 << 30)
    th = threading.Thread(target=work)
    th.start()
    th.join()
    return box["r"]


def _child_json(extra, timeout):
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"{' '.join(extra)}: rc {r.returncode}: {r.stderr.strip()[-300:]}")
    return json.loads(lines[-1])


def cpu_baseline(workload):
    """THE cpu_baseline of the line: the plain-C oracle (kind "port": the CPU restatement that is bit-identical to the flang
    build of the reference, tests/test_oracle_vs_reference.py) on the bench's own grid and state, one core, in a child process
    (a second copy of the 50 GB state: freed when the child ends)."""
    d = _child_json(["--cpu-sample", "--full-grid", "--workload", workload], 900)
    return {"value": d["value"], "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"{d['what']} = the bench's own grid and initial state (identical inputs), internal step 2 of the plain-C oracle "
                      f"(gcc -O2, one core; step 1 untimed: it skips the 3-D body, advance.f:362), {d['seconds']:.1f} s"}


def cpu_baseline_sample(workload, reference):
    """labelled extras: the 1/8 x 1/8 sample (a 64 times smaller working set: friendlier to the caches than the bench grid),
    with the oracle and -- where its build is present -- with the reference's own flang-built sources"""
    from oracle.refharness import have_ref
    case, sim, sjm, kb = _sample_dims(workload)
    if reference and not have_ref(sim, sjm, kb):
        return None
    d = _child_json(["--cpu-sample", "--workload", workload] + (["--reference"] if reference else []), 240)
    who = ("the reference's own sources (solver.f advance.f bounds_forcing.f, AMD flang -O2; linked with input hooks for the PnetCDF "
           "readers this image lacks, oracle/ref_traps.c)") if reference else "the plain-C oracle (gcc -O2)"
    _, im, jm, _, _ = WORKLOADS[workload]
    return {"value": d["value"], "unit": "cell-updates/s", "cores": 1, "kind": "reference" if reference else "port",
            "sample": f"SAMPLE {d['what']} (1/{round(im * jm / (sim * sjm))} of the bench grid's area: a working set that much friendlier to the caches), "
                      f"{d['n']} internal steps of {who}, {d['seconds']:.1f} s"}


def cpu_baseline_all_cores(workload):
    """every host core of the box's CPU share at once, as the reference would use them: the bench's own grid split into one
    whole-row tile per core by the reference's own decomposition arithmetic (extpom_amd/decomp.py = distribute_mpi,
    parallel_mpi.f:34-122), every exchange2d/3d_mpi a real message between the ranks (gloo).  The ranks are child processes
    of this script (`--cpu-tiles-worker`)."""
    import socket
    import subprocess
    case, im, jm, kb, _ = WORKLOADS[workload]
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))        # a 1-GPU box's CPU share
    while cores > 1 and (jm - 2) // cores < 8:
        cores -= 1
    if cores < 2:
        return None
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(cores):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(cores), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   POM_TILE_GRID=f"1x{cores}")
        for k in list(env):
            if k.startswith("TORCHELASTIC_"):
                env.pop(k)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-tiles-worker", "--workload", workload],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    out0 = None
    ok = True
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=600)
            ok = ok and p.returncode == 0
            if r == 0:
                out0 = out
        except Exception:                                          # noqa: BLE001
            p.kill()
            ok = False
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if not ok or not lines:
        return None
    d = json.loads(lines[-1])
    return {"value": d["value"], "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{d['what']} split into 1x{cores} whole-row tiles (one process per core, the reference's distribute_mpi arithmetic, "
                      f"{d['rounds_per_step']:.0f} halo exchanges per step as gloo messages), internal steps 2-{1 + d['n']} of the plain-C oracle, {d['seconds']:.1f} s"}


def _cpu_tiles_worker(workload):
    """one rank of cpu_baseline_all_cores"""
    import torch.distributed as dist
    from extpom_amd import dist as pdist
    from extpom_amd.cases import finish_initial
    from extpom_amd.halo import Halo
    from oracle.pyoracle import OracleTile
    import torch
    torch.set_num_threads(1)
    rank, world, _ = pdist.init("gloo")
    case, im, jm, kb, _ = WORKLOADS[workload]
    tile = pdist.tile_for_rank(rank, world, im, jm)
    st = build_state(workload, tile)
    halo = Halo(tile)
    ot = OracleTile(st, exch2d=halo.numpy_hook2d(), exch3d=halo.numpy_hook3d(), order=halo.numpy_order_hook())
    finish_initial(st, lambda s, a, b, c: ot.call("dens", ot.a3(a), ot.a3(b), ot.a3(c)),
                   lambda s: ot.call("baropg_mcc" if int(s.npg) == 2 else "baropg"))
    ot.run(1)
    pdist.cpu_barrier()                                          # not dist.barrier(): that one opens the GPU (extpom_amd/dist.py)
    c0 = halo.count
    t0 = time.perf_counter()
    n = 0
    while True:
        ot.run(1)
        n += 1
        flag = torch.tensor([1.0 if time.perf_counter() - t0 < 12.0 else 0.0])
        dist.broadcast(flag, 0)                                  # rank 0's clock decides for everybody whether another step follows
        if n >= 6 or (n >= 2 and flag.item() == 0.0):
            break
    pdist.cpu_barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"value": im * jm * kb * n / dt, "what": f"{case} {im}x{jm}x{kb}", "n": n, "seconds": dt,
                          "rounds_per_step": (halo.count - c0) / n}))
    if os.environ.get("POM_BENCH_REPORT_DEVICE_FDS"):            # tests: a CPU worker must never have opened the GPU (the box's process guard counts them)
        fds = set()
        for f in os.listdir("/proc/self/fd"):
            try:
                t = os.readlink(f"/proc/self/fd/{f}")
            except OSError:
                continue
            if "kfd" in t or "/dri/" in t:
                fds.add(t)
        print(f"DEVICE-FDS rank {rank}: {sorted(fds)}", flush=True)
    dist.destroy_process_group()


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
    """`python bench.py --gpus N` without a launcher: start the N rank supervisors as child processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* as torch.distributed.run would set them, rendezvous on 127.0.0.1), pass rank 0's line through and return the worst
    exit code.  This process never initialises the GPU, nothing is re-executed in place -- and no torch.distributed.run stands
    between them: its elastic agent holds the GPU open (tools/diag/ancestors_kfd.py), one process more than the ranks on boxes
    that count them.  (Under the driver's own torch.distributed.run the supervisors are its ranks: bench.supervise either way.)"""
    import socket
    import subprocess
    import threading
    if os.environ.get("POM_BENCH_REHEARSE") != "1":
        import torch
        have = torch.cuda.device_count()                       # counting devices does not initialise the GPU in this process
        if have < n:
            print(f"bench: --gpus {n} but {have} GPU(s) visible -- one rank per GPU, no measurement", file=sys.stderr)
            return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this pool
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE, text=True, bufsize=1))

    def pump(p):
        # the ranks' stdout line by line: the JSON line to stdout, anything a library wrote there (gloo's "[Gloo] Rank ..." banners in
        # a rehearsal) to stderr, so that stdout holds the one line the contract asks for
        for line in p.stdout:
            t = line.strip()
            is_json = t.startswith("{") and t.endswith("}")
            (sys.stdout if is_json else sys.stderr).write(line)
            (sys.stdout if is_json else sys.stderr).flush()

    readers = [threading.Thread(target=pump, args=(p,), daemon=True) for p in procs]
    for t in readers:
        t.start()
    code, failed_at = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            rc = p.poll()
            if rc not in (None, 0) and failed_at is None:
                failed_at = time.perf_counter()                 # a supervisor that died cannot tell the others: give them a moment, then end them
        if failed_at is not None and time.perf_counter() - failed_at > 30.0:
            for p in procs:
                if p.poll() is None:
                    p.kill()                                    # exactly the processes this launcher started
        time.sleep(0.25)
    for t in readers:
        t.join(timeout=5)
    for p in procs:
        code = max(code, abs(p.returncode or 0))
    return code


# ---- tile grids ----------------------------------------------------------------------------------------------------
def tile_grids(workload, n, asked=None):
    """(primary, alternate) as "AxB" strings (nproc_x x nproc_y, parallel_mpi.f:54-65).  Primary: what --tiles asks for; else
    BASELINE's own words where it names a decomposition (configs[2]: "1024x1024x40 ... 2x4 tile decomposition on 8 GPUs"); else
    decomp.choose_tile_grid (whole rows, 1 x N: measured 3-5 % faster per tile).  Alternate: the other
    of {whole rows, most square}, so that a scaling line for configs[2] / [3] always carries 2x4 beside 1x8."""
    from extpom_amd import decomp
    case, im, jm, kb, _ = WORKLOADS[workload]

    def valid(nx, ny):
        iml, jml = decomp.local_size(im, jm, nx, ny)
        return nx * ny == n and decomp.tile_grid(im, jm, iml, jml) == (nx, ny)

    rows = (1, n) if valid(1, n) else None
    square = None
    for nx in range(1, n + 1):
        if n % nx == 0 and valid(nx, n // nx) and nx <= n // nx and (nx, n // nx) != (1, n):
            square = (nx, n // nx)                              # the most square one with nx <= ny: 2x4 of 8, 2x2 of 4
    if asked:
        nx, ny = (int(v) for v in asked.lower().split("x"))
        if not valid(nx, ny):
            raise SystemExit(f"bench: --tiles {asked} does not split {im}x{jm} into {n} tiles")
        prim = (nx, ny)
    elif workload == "basin1024" and n == 8 and valid(2, 4):
        prim = (2, 4)
    else:
        prim = decomp.choose_tile_grid(n, im, jm)
    alt = rows if prim != rows else square
    if alt == prim:
        alt = None
    f = lambda g: f"{g[0]}x{g[1]}" if g else None
    return f(prim), f(alt)


# ---- the rank supervisor (N > 1) -------------------------------------------------------------------------------------------
def supervise(args):
    """One per rank, started by torch.distributed.run.  Touches no GPU.  Runs the passes (module docstring) as child processes
    and -- on rank 0 -- merges their lines into the ONE line of the contract."""
    import datetime
    import signal
    import socket
    import subprocess
    import tempfile
    import threading
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=900))   # CPU only: the supervisors' own channel
    store = dist.distributed_c10d._get_default_store()
    prim, alt = tile_grids(args.workload, world, args.tiles)
    plan = [dict(tiles=prim, overlap=False), dict(tiles=prim, overlap=True)]
    if alt and not args.no_alternate:
        plan.append(dict(tiles=alt, overlap=None))              # None: whichever of the first two was faster
    if args.only_pass is not None:
        if not 0 <= args.only_pass < len(plan):                 # e.g. pass 2 where no alternate grid exists: say so instead of an IndexError on every rank
            if rank == 0:
                print(f"bench: --only-pass {args.only_pass}: this run has passes 0..{len(plan) - 1} (tiles {prim}" + (f", alternate {alt}" if alt and not args.no_alternate else ", no alternate grid") + ")",
                      file=sys.stderr, flush=True)
            dist.destroy_process_group()
            return 2
        plan = [plan[args.only_pass]]
    results = []
    for k, ps in enumerate(plan):
        ps["index"] = k
        if ps["overlap"] is None:
            done = [r for r in results if r["ok"]]
            ps["overlap"] = bool(done and min(done, key=lambda r: r["ms_per_step"])["overlap"])
        box = [None]
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                box[0] = (sk.getsockname()[1], tempfile.mkdtemp(prefix="pombench_"))
        dist.broadcast_object_list(box, src=0)
        port, tmpd = box[0]
        env = dict(os.environ)
        for name in list(env):
            if name.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC", "GROUP_RANK", "ROLE_")):
                env.pop(name)                                   # the child's rendezvous is its own (rank 0 of the pass hosts the store)
        env.update(RANK=str(rank), LOCAL_RANK=str(local), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   POM_TILE_GRID=ps["tiles"], POM_BENCH_PHASE_FILE=os.path.join(tmpd, f"phase{rank}"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if ps["overlap"]:
            env.pop("POMGPU_NO_OVERLAP", None)
        else:
            env["POMGPU_NO_OVERLAP"] = "1"
        cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + ["--rank-pass", json.dumps(ps)]

        def die_with_parent():                                  # an orphaned pass must not outlive its supervisor
            try:
                import ctypes
                ctypes.CDLL("libc.so.6").prctl(1, signal.SIGKILL)   # PR_SET_PDEATHSIG
            except Exception:                                   # noqa: BLE001
                pass
        t_start = time.perf_counter()
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, preexec_fn=die_with_parent)
        lines = []
        rd = threading.Thread(target=lambda: lines.extend(child.stdout), daemon=True)
        rd.start()
        failkey = f"pass{k}_failed"
        limit = float(os.environ.get("POM_BENCH_PASS_LIMIT", 420.0 + 4.0 * (args.steps + args.warmup)))
        # once a pass has completed, the others are the same work on the same machine: a pass that takes many times as long is hung, and
        # the line -- which already has its number -- should not wait minutes for it (every supervisor computes the same limit: wall_s of
        # a completed pass is broadcast below)
        done_ok = [r["wall_s"] for r in results if r["ok"]]
        if done_ok and "POM_BENCH_PASS_LIMIT" not in os.environ:
            limit = min(limit, max(90.0, 8.0 * max(done_ok)))
        why = None
        while True:
            rc = child.poll()
            if rc is not None:
                if rc != 0:
                    store.add(failkey, 1)
                    why = f"exit code {rc}"
                break
            if store.add(failkey, 0) > 0:                       # another rank's child failed: ours can only wait for it for ever
                why = "ended because another rank's process failed"
            elif time.perf_counter() - t_start > limit:
                store.add(failkey, 1)
                why = f"no result within {limit:.0f} s"
            if why:
                child.kill()                                    # exactly the process this supervisor started
                child.wait()
                break
            time.sleep(0.25)
        rd.join(timeout=5)
        phase = None
        try:
            phase = open(env["POM_BENCH_PHASE_FILE"]).read().strip()
        except OSError:
            pass
        mine = {"rank": rank, "rc": child.returncode, "why": why, "phase": phase}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        line = None
        for l in lines:
            t = l.strip()
            if t.startswith("{") and t.endswith("}"):
                line = json.loads(t)
            elif t:
                print(t, file=sys.stderr)
        ok = all(e["rc"] == 0 for e in everyone) and (rank != 0 or line is not None)
        okbox = [ok]
        dist.broadcast_object_list(okbox, src=0)
        ok = okbox[0]
        res = {"tiles": ps["tiles"], "overlap": ps["overlap"], "ok": ok, "wall_s": round(time.perf_counter() - t_start, 1)}
        if ok and rank == 0:
            res.update(ms_per_step=line["ms_per_step"], value=line["value"], line=line)
        if not ok:
            whys = [e["why"] for e in everyone if e["why"]]
            first = ([w for w in whys if w.startswith("exit code")] or [w for w in whys if w.startswith("no result")] or whys or ["unknown"])[0]
            res["failed"] = {"ranks": [e["rank"] for e in everyone if e["rc"] != 0], "first": first, "why_by_rank": {e["rank"]: e["why"] for e in everyone}, "phase_by_rank": {e["rank"]: e["phase"] for e in everyone}}
            if rank == 0:
                print(f"bench: pass {k} (tiles {ps['tiles']}, overlap {'on' if ps['overlap'] else 'off'}) did not complete: {res['failed']}", file=sys.stderr, flush=True)
        if ok and rank != 0:
            res.update(ms_per_step=0.0, value=0.0)
        results.append(res)
        if rank == 0:
            import shutil
            shutil.rmtree(tmpd, ignore_errors=True)             # the pass's phase files
        msbox = [res.get("ms_per_step"), res["wall_s"]]
        dist.broadcast_object_list(msbox, src=0)                # every supervisor plans the next pass from the same numbers
        res["ms_per_step"], res["wall_s"] = msbox
    code = 0
    if rank == 0:
        good = [r for r in results if r["ok"]]
        first = [r for r in good if r["tiles"] == prim] or good
        if not first:
            print("bench: no pass completed -- no measurement", file=sys.stderr, flush=True)
            code = 6
        else:
            best = min(first, key=lambda r: r["ms_per_step"])
            out = best["line"]
            out["config"]["tiles_primary"] = prim
            out["config"]["tiles_note"] = ("primary grid: " + ("--tiles" if args.tiles else "BASELINE configs[2] names 2x4" if (args.workload == "basin1024" and world == 8 and prim == "2x4")
                                           else "whole rows (1xN), the faster split per tile: DESIGN.md section 7") + "; `passes` carries the alternate grid beside it")
            keep = ("ms_per_step", "value", "device_ms_per_step", "kernel_ms_sum_rank0", "message_rounds_ms_rank0", "message_rounds_side_stream_ms_rank0")
            out["passes"] = []
            for r in results:
                e = {"tiles": r["tiles"], "overlap": r["overlap"], "ok": r["ok"], "wall_s": r["wall_s"]}
                if r["ok"]:
                    ln = r["line"]
                    e.update({q: ln.get(q) for q in keep})
                    e.update(rccl_nranks=ln["config"].get("rccl_nranks"), message_rounds_per_step=ln["config"].get("message_rounds_per_step"),
                             message_rounds_per_step_on_side_stream=ln["config"].get("message_rounds_per_step_on_side_stream"), tile=ln["config"].get("tile"),
                             exchange=ln["config"].get("exchange"), primary=(r is best))
                else:
                    e["failed"] = r["failed"]
                out["passes"].append(e)
            print(json.dumps(out), flush=True)
    cbox = [code]
    dist.broadcast_object_list(cbox, src=0)                     # every supervisor leaves with rank 0's verdict
    from extpom_amd.dist import cpu_barrier
    cpu_barrier()                                               # (dist.barrier() would open the GPU in a supervisor)
    dist.destroy_process_group()
    return cbox[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("POM_BENCH_WORKLOAD", "basin2048"), choices=sorted(WORKLOADS))
    ap.add_argument("--tiles", default=None, help="N > 1: the primary tile grid AxB = nproc_x x nproc_y (default: whole rows 1xN; 2x4 for basin1024 on 8 "
                                                  "GPUs, as BASELINE configs[2] names it); the alternate grid is measured beside it")
    ap.add_argument("--no-alternate", action="store_true", help="N > 1: skip the pass on the alternate tile grid")
    ap.add_argument("--only-pass", type=int, default=None, help="N > 1, developer: run only pass 0 (overlap off), 1 (overlap on) or 2 (alternate grid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tune-placement", action="store_true", help="N = 1: leave the 3-D arrays where the allocator put them (pomgpu_tune_placement is not called)")
    ap.add_argument("--side-config", action="store_true",
                    help="N=1: also time BASELINE configs[1] (seamount 256x256x30) in the same run; off by default so that a profile "
                         "of the default command holds the headline workload's kernels only")
    ap.add_argument("--reference", action="store_true", help="internal: with --cpu-sample, time oracle/_ref instead of the oracle")
    ap.add_argument("--full-grid", action="store_true", help="internal: with --cpu-sample, the bench's own grid instead of the 1/8 x 1/8 sample")
    ap.add_argument("--cpu-sample", action="store_true", help="internal: run the one-core oracle leg and print it (no GPU)")
    ap.add_argument("--cpu-tiles-worker", action="store_true", help="internal: one rank of the all-cores oracle leg (no GPU)")
    ap.add_argument("--rank-pass", default=None, help="internal: this process is one rank of one pass (started by its supervisor)")
    ap.add_argument("--profile-all", action="store_true", help="bracket every kernel with events in the timed region")
    ap.add_argument("--storage", choices=["f64", "f32"], default="f64",
                    help="f32: BASELINE configs[4]'s STUDY variant (libpomgpu_f32.so: 3-D arrays stored as fp32, arithmetic and the external "
                         "mode fp64).  Not a parity path (DESIGN.md section 8) and never the headline: the line says so in `dtype`")
    args = ap.parse_args()
    if args.cpu_sample:
        v, what, n, dt = _reference_sample(args.workload) if args.reference else _cpu_run(args.workload, args.full_grid)
        print(json.dumps({"value": v, "what": what, "n": n, "seconds": dt}))
        return
    if args.cpu_tiles_worker:
        _cpu_tiles_worker(args.workload)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))                      # nothing in this process has touched the GPU
    if args.gpus > 1 and args.rank_pass is None:
        sys.exit(supervise(args))                              # ... nor does a rank's supervisor
    measure(args)


def _selftest_rank(args, this_pass, plan):
    """POM_BENCH_SELFTEST='{"hang": [pass, rank], "fail": [pass, rank]}' (tests/test_host_logic.py, no GPU): a stand-in for one
    rank of one pass that joins its pass's rendezvous and then completes, hangs or dies as told -- what the supervisors make of
    it is what is tested.  Its line says SELFTEST in `metric`: it can never pass for a measurement."""
    import torch
    from extpom_amd import dist as pdist
    rank, world, _ = pdist.init("gloo")
    k = this_pass["index"]
    pf = os.environ.get("POM_BENCH_PHASE_FILE")
    if pf:
        open(pf, "w").write("connect")
    if plan.get("fail") == [k, rank]:
        sys.exit(9)
    if plan.get("hang") == [k, rank]:
        time.sleep(3600)
    pdist.cpu_barrier()
    if rank == 0:
        ms = 10.0 + k - (3.0 if this_pass["overlap"] else 0.0)
        print(json.dumps({"metric": "SELFTEST -- not a measurement", "value": 1.0 / ms, "ms_per_step": ms, "n_gpus": world, "steps": args.steps,
                          "config": {"tiles": os.environ.get("POM_TILE_GRID"), "overlap": this_pass["overlap"], "rccl_nranks": 0,
                                     "no_overlap_env": os.environ.get("POMGPU_NO_OVERLAP")}}), flush=True)
    pdist.cpu_barrier()
    torch.distributed.destroy_process_group()


def measure(args):
    """the measuring process: N = 1, or one rank of one pass"""
    this_pass = json.loads(args.rank_pass) if args.rank_pass else None
    if this_pass is not None and os.environ.get("POM_BENCH_SELFTEST"):
        return _selftest_rank(args, this_pass, json.loads(os.environ["POM_BENCH_SELFTEST"]))
    import torch
    from extpom_amd import decomp, dist as pdist
    # POM_BENCH_REHEARSE=1: developer rehearsal of the N > 1 code path on a ONE-GPU box -- every rank on GPU 0,
    # gloo with host-staged halos instead of RCCL (RCCL refuses two ranks on one device).  Not a measurement.
    rehearse = os.environ.get("POM_BENCH_REHEARSE") == "1"
    # the ranks' CONTROL plane (barriers, the clock's maximum, the RCCL unique id) is a gloo group on the CPU: the only RCCL
    # communicators of the process are the library's own (data path), so no two communicators ever meet on the GPU by accident
    rank, world, local = pdist.init("gloo")
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
    from extpom_amd import lib as _L
    g = gpu_initialise(st, local, stream, _L.LIBPATH_F32 if f32 else None)
    # N = 1: where in HBM the 3-D arrays start is measured, not taken as it comes (pomgpu_tune_placement, include/pomgpu.h: up to 6 % per
    # kernel, different from process to process, reproducible inside one): a few start offsets inside one allocation, three real steps
    # each, the fastest kept -- before the warm-up; the line says what was tried (config.placement).  --no-tune-placement skips it.
    placement = None
    if world == 1 and not args.no_tune_placement:
        try:
            g.run(2)                                                # (the model's first step skips its 3-D part, advance.f:362: not a step to time)
            placement = g.tune_placement(3, 10)
        except Exception as e:                                     # noqa: BLE001 -- the measurement does not depend on it
            print(f"bench: placement not tuned ({e})", file=sys.stderr)
    build_id = g.L.pomgpu_build_id().decode()
    exchange = "none"
    rccl_nranks = 0
    # N > 1: a rank that is lost, or a message round whose partner never posts, would leave the others waiting inside
    # RCCL for ever.  Every phase -- connecting, the first steps (RCCL sets its channels up lazily), the timed steps --
    # gets a deadline; a rank that misses one says where it was and leaves with a non-zero code, which ends the pass
    # (its supervisor tells the other ranks' supervisors, bench.supervise).
    phase = ["start", None]
    phase_file = os.environ.get("POM_BENCH_PHASE_FILE")

    def deadline(name, seconds):
        import threading
        if phase[1] is not None:
            phase[1].cancel()
            phase[1] = None
        phase[0] = name
        if phase_file:
            try:
                with open(phase_file, "w") as f:
                    f.write(name)
            except OSError:
                pass
        if world > 1 and seconds:
            def expired():
                print(f"bench[{rank}]: '{name}' did not complete within {seconds:.0f} s -- giving up", file=sys.stderr, flush=True)
                os._exit(4)
            phase[1] = threading.Timer(seconds, expired)
            phase[1].daemon = True
            phase[1].start()

    deadline("connect", 180.0)
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
                exchange = "library exchange, native RCCL send/recv on the kernels' stream"
                rccl_nranks = g.rccl_nranks()
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
    deadline("warm-up steps", 180.0)
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
    deadline("wrap-up", 180.0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    g.get_con()
    err = int(st.error_status)
    if world > 1:                                              # a rank whose step failed fails the pass: every rank learns it
        t = torch.tensor([float(err)], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        err = int(t.item())

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
        overlap_on = bool(this_pass and this_pass.get("overlap"))
        out = {
            "metric": "3D cell-updates/sec on internal mode (one internal step = the 3-D baroclinic step and its isplit 2-D external substeps; whole-step wall time)",
            "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32-storage of the 3-D arrays / f64 arithmetic and external mode (STUDY variant, not a parity path)" if f32 else "f64",
            "data": "synthetic",
            "config": {"workload": desc + f", mode=3 nadv=2 nitera=1 npg=1 dte=6 isplit=30", "tiles": f"{tile.nproc_x}x{tile.nproc_y}",
                       "tile": f"{tile.im_local}x{tile.jm_local}x{kb}", "global_cells": cells, "exchange": exchange,
                       "overlap": (("second stream + second communicator: nine of a step's ten message rounds (the early part of the wide exchange, advct's edge lines, advx + advy + aam, w, the turbulence arrays, T / S / rho, the two velocity exchanges that end mode_internal, wr); between kernels: the late part of the wide exchange" if overlap_on
                                    else "off: every message round on the kernels' stream (POMGPU_NO_OVERLAP)") if world > 1 else None),
                       # N = 1: start offsets of blk3d inside its allocation that were tried (MiB), ms per step of each, which was kept.  N > 1: the tiles run where
                       # the allocator put them ("untuned": every rank would have to try the same number of layouts for their message rounds to match, and an 8-tile
                       # gains nothing) -- a scaling efficiency is therefore read against n1_untuned_ms_per_step of the N = 1 line, like against like
                       "placement": placement if world == 1 else "untuned",
                       "rccl_nranks": rccl_nranks,                  # what ncclCommCount reports for the library's communicator (0: no RCCL transport)
                       "message_rounds_ms_rank0": round(msg_ms, 3),
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
            # N = 1: the step as the allocator placed the arrays -- the tuner's first trial is that layout (three steps, events on the kernels' stream); what an
            # N > 1 line, whose tiles are not tuned, should be compared with
            "n1_untuned_ms_per_step": (placement["ms_per_step"][0] if (world == 1 and placement and placement.get("tried")) else (round(ms, 3) if world == 1 else None)),
            # what ONE tile of the N-tile split costs outside the transfers, measured on one GPU through the multi-tile code path (tools/tile_probe.py, committed
            # under profiles/): the efficiency a measured scaling curve can at most reach, to be read beside it
            "efficiency_bound_from_tile_probe": _tile_probe_bounds(args.workload),
            "host_libc": _libc_version(),     # bit-parity with the reference leans on this libm's pow (THIRD_PARTY_NOTICES.md)
        }
        if not args.no_cpu_baseline and world == 1:
            # the oracle on the bench's own grid (kind "port") is THE cpu_baseline; the sample figures are labelled extras
            free_state = args.workload in ("basin2048", "basin1024")
            if free_state and not args.side_config:
                g.close()                                         # the CPU legs need the host's memory, not the GPU's
                g = None
                st = None
            for name, fn in (("cpu_baseline", lambda: cpu_baseline(args.workload)),
                             ("cpu_baseline_all_cores", lambda: cpu_baseline_all_cores(args.workload)),
                             ("cpu_baseline_sample_reference", lambda: cpu_baseline_sample(args.workload, True)),
                             ("cpu_baseline_sample_port", lambda: cpu_baseline_sample(args.workload, False))):
                try:
                    r = fn()
                    if r:
                        out[name] = r
                except Exception as e:                             # noqa: BLE001 -- a CPU leg must not cost the GPU measurement its line
                    print(f"bench: {name} unavailable ({e})", file=sys.stderr)
            if "cpu_baseline" not in out and "cpu_baseline_sample_port" in out:
                out["cpu_baseline"] = out["cpu_baseline_sample_port"]
        if world == 1 and args.workload == "basin2048" and args.side_config:
            # BASELINE configs[1] (seamount 256x256x30) in the same run: a launch-latency-bound size, reported beside
            # the headline grid so that both ends of the size range are on the line
            if g is not None:
                g.close()
            g = None
            out["config1_seamount256"] = side_config("seamount256", local, stream)
        print(json.dumps(out), flush=True)
    if g is not None:
        g.close()
    if os.environ.get("POM_BENCH_REPORT_ANCESTORS") and rank == 0:   # developer: which of launcher / elastic agent / supervisor hold the GPU open?
        import runpy
        runpy.run_path(os.path.join(ROOT, "tools", "diag", "ancestors_kfd.py"))
    if world > 1:
        torch.distributed.barrier()
        deadline("done", 0)
        torch.distributed.destroy_process_group()
    if err:
        sys.exit(7)                                            # error_status = 1 on some rank: the pass did not measure a valid run


if __name__ == "__main__":
    main()
