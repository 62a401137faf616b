"""developer tool (GPU box): per-kernel times of several library variants measured INTERLEAVED in one process.
Run-to-run and box-to-box differences on the pool are +-5 % (clocks); variants timed in separate processes cannot be
ranked closer than that.  Here the state is built once, every variant gets its own context (60 GB each at the default
size) and the variants take turns: R rounds of (profiled step of A, of B, ...); reported: min and median per kernel.
usage: python tools/kbench.py [--workload W] [--rounds R] [--kernels a,b,...] tag[=lib.so|=@|=@lib.so][:ENV=V ...] ...   (=@: the previous variant's context; =@lib.so: that context under another build)
(an ENV of a spec is set while that variant's context is created and while it runs)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from extpom_amd import dist as pdist
from extpom_amd.model import PomGpu
from extpom_amd import lib as _lib
_lib._SIGS.pop("pomgpu_rccl_available", None)     # older variant libraries do not export it

args = sys.argv[1:]
wl, rounds, kernels = "basin2048", 5, None
while args and args[0].startswith("--"):
    if args[0] == "--workload": wl = args[1]
    elif args[0] == "--rounds": rounds = int(args[1])
    elif args[0] == "--kernels": kernels = args[1].split(",")
    args = args[2:]
case, im, jm, kb, _ = bench.WORKLOADS[wl]
tile = pdist.tile_for_rank(0, 1, im, jm)
st0 = bench.build_state(wl, tile)
g0 = bench.gpu_initialise(st0, 0, None)      # finishes the initial state with the default library
if not os.environ.get('KBENCH_KEEP_G0'): g0.close()
variants = []
for spec in args:
    parts = spec.split(":")
    tag, _, lib = parts[0].partition("=")
    env = dict(p.split("=", 1) for p in parts[1:])
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    if lib == "@":                  # same context as the previous variant: only the switches differ (set on the live context per turn)
        g = variants[-1][2]
    elif lib.startswith("@"):       # the previous variant's context (its memory) driven by ANOTHER build of the library: the handle is a
        import copy                 # plain C struct of the same layout -- for variants that differ in kernel code only
        g = copy.copy(variants[-1][2])
        g.L = _lib.load(os.path.join(ROOT, lib[1:]))
    else:
        g = PomGpu(st0, device=0, libpath=os.path.join(ROOT, lib) if lib else None)
    for k, v in env.items():
        if k.startswith("POMGPU_") and k != "POMGPU_LIBPATH": g.switch(k, v)
    g.run(2); g.sync()
    for k in env:
        if k.startswith("POMGPU_") and k != "POMGPU_LIBPATH": g.switch(k, None)
    for k, v in old.items():
        os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    variants.append((tag, env, g, {}))
    b3, b2 = g.device_ptr("aam"), g.device_ptr(list(__import__("extpom_amd.layout", fromlist=["P2"]).P2)[0])
    print(f"{tag}: blk3d at 0x{b3:x} (mod 2MiB {b3 % (2 << 20)}, mod 1GiB {(b3 % (1 << 30)) >> 20} MiB), blk2d at 0x{b2:x}", flush=True)
for r in range(rounds):
    for tag, env, g, acc in variants:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        for k, v in env.items():            # the library reads its switches at pomgpu_create: a variant's are set on the live context for its turn
            if k.startswith("POMGPU_") and k != "POMGPU_LIBPATH": g.switch(k, v)
        g.prof_begin(); g.run(1); prof = g.prof_end()
        for k in env:
            if k.startswith("POMGPU_") and k != "POMGPU_LIBPATH": g.switch(k, None)
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        prof.pop("msg_round", None)
        prof.pop("phase_step", None); prof.pop("phase_external", None)      # brackets around other brackets (pomgpu.h): not kernels
        ext = ("k_ext_", "k_advave_", "k_modeint_tail", "k_int_tail", "k_check_velocity", "k_check_areas", "k_copy2", "k_bcond1")
        acc.setdefault("internal", []).append(sum(v[1] for k, v in prof.items() if not k.startswith(ext)))
        acc.setdefault("external", []).append(sum(v[1] for k, v in prof.items() if k.startswith(ext)))
        for k, v in prof.items():
            acc.setdefault(k[2:], []).append(v[1])
names = ["internal", "external"] + sorted((k for k in variants[0][3] if k not in ("internal", "external") and (kernels is None or k in kernels)),
                                          key=lambda k: -min(variants[0][3][k]))
if kernels is None:
    names = names[:22]
print(f"{'min / median (ms)':22s}" + "".join(f"{t[0]:>18s}" for t in variants))
for n in names:
    print(f"{n:22s}" + "".join(f"{min(t[3].get(n, [0])):9.3f}/{statistics.median(t[3].get(n, [0])):7.3f} " for t in variants))
sys.stdout.flush()
os._exit(0)      # contexts shared between builds (=@lib.so) must not be destroyed twice at interpreter exit
