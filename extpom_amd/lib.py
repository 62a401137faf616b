"""ctypes binding of the C ABI (include/pomgpu.h) -- loads extpom_amd/csrc/libpomgpu.so.

There is NO CPU fallback: if the shared library is missing or no HIP device can be opened the
import / context creation fails loudly (PomGpuError).
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, "csrc", "libpomgpu.so")
# the study variant of BASELINE configs[4]: 3-D arrays stored as fp32, arithmetic and the 2-D external mode fp64
# (same sources, -DPOMGPU_STORE_F32); never the default, never loaded unless asked for by path
LIBPATH_F32 = os.path.join(HERE, "csrc", "libpomgpu_f32.so")


class PomGpuError(RuntimeError):
    pass


class FileMeta(ctypes.Structure):
    """pomgpu_file_meta (include/pomgpu.h)"""
    _fields_ = [("title", ctypes.c_char_p), ("time_start", ctypes.c_char_p), ("im_global", ctypes.c_int), ("jm_global", ctypes.c_int),
                ("i0", ctypes.c_int), ("j0", ctypes.c_int), ("create", ctypes.c_int), ("stats", ctypes.POINTER(ctypes.c_double))]


class Dims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("im", "jm", "kb", "im_local", "jm_local", "n_west", "n_east", "n_south", "n_north")]


EXCHANGE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                               ctypes.c_int)
# pomgpu_order_fn: (user, send_east, n_east, send_north, n_north, recv_west, recv_south) -- device addresses
ORDER_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                            ctypes.c_void_p, ctypes.c_void_p)

# pomgpu_transport_fn: (user, send[8], scount[8], recv[8], rcount[8]) -- device addresses, counts in doubles
TRANSPORT_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t))
# direction order of every eight-neighbour table of the C ABI, and the direction a message arrives from
DIRS = ("w", "e", "s", "n", "sw", "se", "nw", "ne")
OPP = (1, 0, 3, 2, 7, 6, 5, 4)

_P = ctypes.c_void_p
_I = ctypes.c_int
_SIGS = {
    "pomgpu_create": (_I, [ctypes.POINTER(_P), ctypes.POINTER(Dims), _I, _P]),
    "pomgpu_destroy": (None, [_P]),
    "pomgpu_last_error": (ctypes.c_char_p, [_P]),
    "pomgpu_sync": (_I, [_P]),
    "pomgpu_stream": (_P, [_P]),
    "pomgpu_current_stream": (_P, [_P]),
    "pomgpu_upload": (_I, [_P, _P, _P, _P, _P, _P, _I]),
    "pomgpu_download": (_I, [_P, _P, _P, _P, _P, _P]),
    "pomgpu_upload_2d": (_I, [_P, _I, _P]),
    "pomgpu_upload_3d": (_I, [_P, _I, _P]),
    "pomgpu_download_2d": (_I, [_P, _I, _P]),
    "pomgpu_download_3d": (_I, [_P, _I, _P]),
    "pomgpu_set_con": (_I, [_P, _P, _I]),
    "pomgpu_get_con": (_I, [_P, _P]),
    "pomgpu_bind_host": (_I, [_P, _P, _P]),
    "pomgpu_set_restore_record": (_I, [_P, _I, _P, _P]),
    "pomgpu_set_forcing_record": (_I, [_P, _I, _I, _P, _P]),
    "pomgpu_set_lateral_record": (_I, [_P, _I, ctypes.POINTER(ctypes.c_void_p)]),
    "pomgpu_device_2d": (_P, [_P, _I]),
    "pomgpu_device_3d": (_P, [_P, _I]),
    "pomgpu_set_exchange": (_I, [_P, EXCHANGE_FN, _P]),
    "pomgpu_set_order_exchange": (_I, [_P, ORDER_FN, _P]),
    "pomgpu_set_transport": (_I, [_P, ctypes.POINTER(_I), TRANSPORT_FN, _P]),
    "pomgpu_rccl_available": (_I, [ctypes.c_char_p]),
    "pomgpu_rccl_unique_id": (_I, [_P, ctypes.c_char_p]),
    "pomgpu_rccl_init": (_I, [_P, _P, _I, _I, ctypes.POINTER(_I), ctypes.c_char_p]),
    "pomgpu_transport_side_capable": (_I, [_P]),
    "pomgpu_transport_side_agree": (_I, [_P, _I]),
    "pomgpu_transport_stream_ordered": (_I, [_P, _I]),
    "pomgpu_switch_digest": (ctypes.c_uint, [_P]),
    "pomgpu_rccl_nranks": (_I, [_P]),
    "pomgpu_debug_switch": (_I, [_P, ctypes.c_char_p, ctypes.c_char_p]),
    "pomgpu_exchange_rounds": (ctypes.c_long, [_P]),
    "pomgpu_exchange_rounds_side": (ctypes.c_long, [_P]),
    "pomgpu_set_wide_external": (_I, [_P, _I, _I, _I]),
    "pomgpu_halo_pack": (_I, [_P, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_I), _I, _I, _P, _P]),
    "pomgpu_halo_unpack": (_I, [_P, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_I), _I, _I, _P, _P]),
    "pomgpu_halo_pack8": (_I, [_P, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_I), _I, ctypes.POINTER(ctypes.c_void_p)]),
    "pomgpu_halo_unpack8": (_I, [_P, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_I), _I, ctypes.POINTER(ctypes.c_void_p)]),
    "pomgpu_check_velocity": (_I, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "pomgpu_run": (_I, [_P, _I]),
    "pomgpu_io_wait": (_I, [_P]),
    "pomgpu_write_output": (_I, [_P, ctypes.c_char_p, ctypes.POINTER(FileMeta)]),
    "pomgpu_write_restart": (_I, [_P, ctypes.c_char_p, ctypes.POINTER(FileMeta)]),
    "pomgpu_domain_stats": (_I, [_P, ctypes.POINTER(ctypes.c_double), _I]),
    "pomgpu_advq": (_I, [_P, _P, _P, _P]),
    "pomgpu_advt1": (_I, [_P, _P, _P, _P, _P]),
    "pomgpu_advt2": (_I, [_P, _P, _P, _P, _P]),
    "pomgpu_dens": (_I, [_P, _P, _P, _P]),
    "pomgpu_proft": (_I, [_P, _P, _P, _P, _I]),
    "pomgpu_bcond": (_I, [_P, _I]),
    "pomgpu_bcondorl": (_I, [_P, _I]),
    "pomgpu_prof_begin": (_I, [_P]),
    "pomgpu_prof_filter": (_I, [_P, ctypes.c_char_p]),
    "pomgpu_prof_end": (_I, [_P]),
    "pomgpu_prof_count": (_I, [_P]),
    "pomgpu_prof_get": (_I, [_P, _I, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_long),
                             ctypes.POINTER(ctypes.c_double)]),
    "pomgpu_version": (ctypes.c_char_p, []),
    "pomgpu_build_id": (ctypes.c_char_p, []),
    "pomgpu_tune_placement": (_I, [_P, _I, _I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(_I), ctypes.POINTER(_I)]),
}
# argument-less hot-path entry points, same names as the reference subroutines
NOARG = ["get_time", "lateral_viscosity", "mode_interaction", "mode_external", "mode_internal", "advance", "advave",
         "advct", "advu", "advv", "baropg", "baropg_mcc", "wind", "heat", "surface", "surface_forcing", "lateral_bc", "profq", "profu", "profv", "vertvl", "realvertvl", "restore_interior"]
for _n in NOARG:
    _SIGS["pomgpu_" + _n] = (_I, [_P])

EXPORTS = sorted(_SIGS)

_cache = {}


def load(path: str | None = None):
    path = path or os.environ.get("POMGPU_LIBPATH") or LIBPATH     # POMGPU_LIBPATH: developer builds (kernel variants)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise PomGpuError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the hot path)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)       # AttributeError = the library does not export the C ABI
        fn.restype = res
        fn.argtypes = args
    _cache[path] = lib
    return lib
