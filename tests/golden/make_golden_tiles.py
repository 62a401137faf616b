"""Generates tests/golden/tiles_65x49x21_2x2.json by running the REFERENCE ITSELF on four MPI ranks:

    oracle/build_ref.sh 65 49 21 34 26 4
    /opt/conda/bin/mpiexec -n 4 python tests/golden/make_golden_tiles.py

Every rank loads oracle/_ref/libpomref_65x49x21_34x26p4.so (the unmodified solver.f advance.f bounds_forcing.f
initialize.f parallel_mpi.f, n_proc = 4, im_local x jm_local = 34 x 26), calls the reference's own initialize_mpi and
distribute_mpi (parallel_mpi.f:6-122), takes its tile of the global case, and steps it with the reference's routines --
every exchange2d_mpi / exchange3d_mpi / order2d_mpi / order3d_mpi (parallel_mpi.f:154-480) a real MPICH message
between the four processes.  Recorded per rank: what distribute_mpi put into blkpar / blksiz (my_task, im, jm, i_global,
j_global, the four neighbours) and the SHA-256 of every restart-list field over the tile's (jm, im) cells, ghost cells
INCLUDED, at several steps, for npg = 1 and npg = 2.  tests/test_oracle_golden.py holds extpom_amd.decomp against the
first and the oracle + extpom_amd.halo tiles against the second; tests/test_gpu_multitile.py the HIP tiles.
"""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from extpom_amd import decomp  # noqa: E402
from extpom_amd.cases import cut_tile, finish_initial, make_case  # noqa: E402
from extpom_amd.layout import RESTART_2D, RESTART_3D  # noqa: E402
from oracle.refharness import RefLib, ref_path  # noqa: E402

IM, JM, KB, IML, JML, NP = 65, 49, 21, 34, 26, 4
CONFIGS = {
    "seamount_2x2": ("seamount", dict(dte=6.0, isplit=30), [1, 2, 3, 10]),
    "seamount_2x2_npg2": ("seamount", dict(dte=6.0, isplit=30, npg=2), [1, 3, 10]),
    "island_2x2": ("island", dict(dte=6.0, isplit=30), [3, 10]),
    "seamount_2x2_isplit10": ("seamount", dict(dte=6.0, isplit=10), [1, 3, 10]),     # tiles wide enough for the wide-halo external mode (w = 14)
    "seamount_2x2_isplit10_npg2": ("seamount", dict(dte=6.0, isplit=10, npg=2), [3, 10]),
}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<f8").tobytes()).hexdigest()


class TileRef(RefLib):
    """RefLib whose put() leaves blkpar alone: my_task, pom_comm and the neighbours are what the reference's own
    initialize_mpi / distribute_mpi made them"""

    def put(self, st):
        keep = self.par.copy()
        super().put(st)
        self.par[:] = keep


def main():
    lib = TileRef(IML, JML, KB, path=ref_path(IM, JM, KB, IML, JML, NP))
    lib.call("initialize_mpi")
    lib.call("distribute_mpi")
    par = lib.par.copy()
    rank = int(par[0])
    siz = {k: int(lib.siz[k][0]) for k in lib.siz.dtype.names}
    info = {"my_task": rank, "master_task": int(par[1]), "siz": siz,
            "i_global": [int(v) for v in par[3:3 + IML]], "j_global": [int(v) for v in par[3 + IML:3 + IML + JML]],
            "n_west": int(par[-4]), "n_east": int(par[-3]), "n_south": int(par[-2]), "n_north": int(par[-1])}
    tile = decomp.make_tile(rank, IM, JM, IML, JML, n_proc=NP)
    out = {"info": info, "configs": {}}
    for name, (case, nml, checkpoints) in CONFIGS.items():
        st = make_case(case, IM, JM, KB, tile=tile, **nml)

        def dens(s, si, ti, rho):
            lib.put(s); lib.call("dens", lib.f3(si), lib.f3(ti), lib.f3(rho)); lib.get(s)

        def baropg(s):
            lib.put(s); lib.call("baropg_mcc" if int(s.npg) == 2 else "baropg"); lib.get(s)

        finish_initial(st, dens, baropg)
        lib.put(st)
        cfg = {"case": case, "nml": nml, "steps": {}}
        jm, im = siz["jm"], siz["im"]
        for n in range(1, max(checkpoints) + 1):
            lib.con["iint"][0] = n
            lib.advance()
            if n in checkpoints:
                lib.get(st)
                cfg["steps"][str(n)] = {f: digest(st.field(f)[..., :jm, :im]) for f in RESTART_2D + RESTART_3D}
        assert int(lib.con["error_status"][0]) == 0
        out["configs"][name] = cfg
    # collect on rank 0 through files (no mpi4py here)
    tmp = os.environ.get("TILES_TMP", "/tmp/tiles_golden")
    os.makedirs(tmp, exist_ok=True)
    with open(os.path.join(tmp, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    lib.lib.mpi_barrier_.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    comm, ierr = ctypes.c_int(int(par[2])), ctypes.c_int(0)
    lib.lib.mpi_barrier_(ctypes.byref(comm), ctypes.byref(ierr))
    if rank == 0:
        allr = [json.load(open(os.path.join(tmp, f"rank{r}.json"))) for r in range(NP)]
        gold = {"grid": [IM, JM, KB], "local": [IML, JML], "n_proc": NP, "fields": RESTART_2D + RESTART_3D,
                "ranks": [a["info"] for a in allr],
                "configs": {name: {"case": allr[0]["configs"][name]["case"], "nml": allr[0]["configs"][name]["nml"],
                                   "steps": {s: [a["configs"][name]["steps"][s] for a in allr] for s in allr[0]["configs"][name]["steps"]}}
                            for name in CONFIGS}}
        with open(os.path.join(ROOT, "tests", "golden", "tiles_65x49x21_2x2.json"), "w") as f:
            json.dump(gold, f, indent=1, sort_keys=True)
        print("wrote tiles_65x49x21_2x2.json", flush=True)
    lib.call("finalize_mpi")


if __name__ == "__main__":
    import threading
    threading.stack_size(1 << 29)
    t = threading.Thread(target=main)
    t.start()
    t.join()
