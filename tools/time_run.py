"""developer tool: wall time of PomGpu.run(n) without any kernel profiling (the path a production run takes)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "seamount256"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
from extpom_amd import dist as pdist
case, im, jm, kb, _ = bench.WORKLOADS[wl]
tile = pdist.tile_for_rank(0, 1, im, jm)
st = bench.build_state(wl, tile)
g = bench.gpu_initialise(st, 0, None)
g.run(5); g.sync()
t0 = time.perf_counter(); g.run(n); g.sync(); dt = time.perf_counter() - t0
print(f"{wl}: {dt / n * 1e3:.3f} ms/step over {n} steps  ({im * jm * kb * n / dt:.3e} cell-updates/s)")
