"""Generates tests/golden/seamount_65x49x21.json (+ .npz planes) and tests/golden/kb50_256x192x50.json by running the REFERENCE itself
(oracle/_ref/libpomref_65x49x21.so, built from the unmodified sources by oracle/build_ref.sh) on
the inputs of extpom_amd.cases.  Run from the repo root in a container that has /root/reference:

    oracle/build_ref.sh 65 49 21 && oracle/build_ref.sh 256 192 50 && python tests/golden/make_golden.py [kb50 | kb50long NAME | forced]

The fixture holds, per configuration and checkpoint step, the SHA-256 of every restart-list field
(the prognostic state, reference io_pnetcdf.F:1724-1886) exactly as the reference left it in its
COMMON blocks, and float64 planes of a few fields for tolerance-based (GPU) comparisons.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from extpom_amd.cases import make_case  # noqa: E402
from extpom_amd.layout import RESTART_2D, RESTART_3D  # noqa: E402
from oracle.refharness import RefLib, ref_finish_initial  # noqa: E402

IM, JM, KB = 65, 49, 21
CONFIGS = {
    # name: (case, namelist overrides, checkpoints)
    "seamount_default": ("seamount", dict(dte=6.0, isplit=30), [1, 2, 3, 10, 100]),
    "seamount_nadv1": ("seamount", dict(dte=6.0, isplit=30, nadv=1), [3, 20]),
    "seamount_nitera2": ("seamount", dict(dte=6.0, isplit=30, nitera=2), [3, 20]),
    "seamount_mode2": ("seamount", dict(dte=6.0, isplit=30, mode=2), [3, 20]),
    "seamount_mode4": ("seamount", dict(dte=6.0, isplit=30, mode=4), [3, 20]),
    "seamount_nbct2": ("seamount", dict(dte=6.0, isplit=30, nbct=2), [3, 10]),
    "seamount_nbc3": ("seamount", dict(dte=6.0, isplit=30, nbct=3, nbcs=3), [3, 20]),
    "island_default": ("island", dict(dte=6.0, isplit=30), [3, 40]),
    "basin_default": ("basin", dict(dte=6.0, isplit=30), [3, 40]),
    "basin_alpha": ("basin", dict(dte=6.0, isplit=10, alpha=0.225), [3, 20]),
    "seamount_npg2": ("seamount", dict(dte=6.0, isplit=30, npg=2), [3, 20]),      # baropg_mcc
    "island_npg2": ("island", dict(dte=6.0, isplit=30, npg=2), [3, 20]),
}
PLANES = {"seamount_default": [10, 100], "island_default": [40], "basin_default": [40]}
PLANE_FIELDS = ["el", "et", "ua", "va", "u", "v", "t", "s", "q2", "km", "rho", "w"]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<f8").tobytes()).hexdigest()


# kb = 50: the level count of the benchmarked grid (2048x1536x50), i.e. the <50> instantiations of the register-resident
# column kernels; reference build oracle/_ref/libpomref_256x192x50.so (oracle/build_ref.sh 256 192 50)
CONFIGS_KB50 = {
    "basin50_default": ("basin", dict(dte=6.0, isplit=30), [1, 3]),
    "basin50_nadv1": ("basin", dict(dte=6.0, isplit=30, nadv=1), [3]),
    "basin50_npg2": ("basin", dict(dte=6.0, isplit=30, npg=2), [3]),
    "seamount50_default": ("seamount", dict(dte=6.0, isplit=30), [3]),
}


# north_star's bar -- 1000 internal steps -- at the benchmark's level count: checkpoints 100 / 500 / 1000 of the reference
# itself at 256x192x50 (about ten minutes of one core per configuration; `kb50long NAME` writes kb50_1000steps_NAME.json)
CONFIGS_KB50_LONG = {
    "basin50_default": ("basin", dict(dte=6.0, isplit=30), [100, 500, 1000]),
    "seamount50_default": ("seamount", dict(dte=6.0, isplit=30), [100, 500, 1000]),
}


# One configuration stepped by the reference's OWN `advance` (advance.f:6-59) instead of the harness's restatement of its
# sequence: surface_forcing and lateral_bc run as the reference calls them (their PnetCDF readers are the input hooks of
# oracle/ref_traps.c, fed the records of extpom_amd.cases), print_section / write_output / write_restart stay silent
# because iprint and irestart lie beyond the run (prtd1 = 0.5 d -> iprint = 120 steps at dti = 360 s).
FORCED = ("seamount", dict(dte=6.0, isplit=60, days=1.0, prtd1=0.5), [1, 2, 10, 11, 30, 31])


def generate_forced():
    from extpom_amd.cases import make_forcing_records, make_lateral_records
    case, nml, checkpoints = FORCED
    st = make_case(case, 65, 49, 21, **nml)
    ref_finish_initial(st)
    make_forcing_records(st, 4)
    make_lateral_records(st, 8)
    lib = RefLib(65, 49, 21)
    lib.mpi_init()
    lib.put(st)
    assert int(lib.con["iprint"][0]) > max(checkpoints) and int(lib.con["irestart"][0]) > max(checkpoints)
    cfg = {"case": case, "nml": nml, "forcing_records": 4, "lateral_records": 8, "steps": {}}
    for n in range(1, max(checkpoints) + 1):
        lib.con["iint"][0] = n
        lib.call("advance")                      # the reference's own subroutine
        if n in checkpoints:
            lib.get(st)
            cfg["steps"][str(n)] = {f: digest(st.field(f)) for f in RESTART_2D + RESTART_3D}
            cfg["steps"][str(n)]["bdry"] = digest(st.bdry)
    assert int(lib.con["error_status"][0]) == 0
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "forced_advance_65x49x21.json"), "w") as f:
        json.dump({"grid": [65, 49, 21], "fields": RESTART_2D + RESTART_3D, "config": cfg}, f, indent=1, sort_keys=True)
    print("forced_advance done", flush=True)


def main():
    if "forced" in sys.argv[1:]:
        return generate_forced()
    if "kb50long" in sys.argv[1:]:
        name = sys.argv[sys.argv.index("kb50long") + 1]
        return generate(256, 192, 50, {name: CONFIGS_KB50_LONG[name]}, {}, "kb50_1000steps_" + name)
    if "kb50" not in sys.argv[1:]:
        generate(65, 49, 21, CONFIGS, PLANES, "seamount_65x49x21")
    generate(256, 192, 50, CONFIGS_KB50, {}, "kb50_256x192x50")


def generate(IM, JM, KB, CONFIGS, PLANES, stem):
    out = {"grid": [IM, JM, KB], "fields": RESTART_2D + RESTART_3D, "configs": {}}
    planes = {}
    for name, (case, nml, checkpoints) in CONFIGS.items():
        st = make_case(case, IM, JM, KB, **nml)
        ref_finish_initial(st)
        cfg = {"case": case, "nml": nml, "init": {f: digest(st.field(f)) for f in RESTART_2D + RESTART_3D},
               "steps": {}}
        lib = RefLib(IM, JM, KB)
        lib.put(st)
        for n in range(1, max(checkpoints) + 1):
            lib.con["iint"][0] = n
            lib.advance()
            if n in checkpoints:
                lib.get(st)
                cfg["steps"][str(n)] = {f: digest(st.field(f)) for f in RESTART_2D + RESTART_3D}
                if n in PLANES.get(name, []):
                    for f in PLANE_FIELDS:
                        a = st.field(f)
                        planes[f"{name}/{n}/{f}"] = (a if a.ndim == 2 else a[[0, KB // 2, KB - 2]]).copy()
        out["configs"][name] = cfg
        print(name, "done", flush=True)
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, stem + ".json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    if planes:
        np.savez_compressed(os.path.join(here, stem + "_planes.npz"), **planes)


if __name__ == "__main__":
    import threading
    threading.stack_size(1 << 30)       # the reference's automatic (im,jm,kb) arrays live on the caller's stack
    t = threading.Thread(target=main)
    t.start()
    t.join()
