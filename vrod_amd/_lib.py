"""ctypes loader for libvrod_hip.so (the C ABI of include/vrod.h).

There is NO Python or CPU fallback: if the shared library is missing or cannot be
loaded this raises, and every entry point fails without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VROD_HIP_LIB: load another build of the same library (A/B timing of two builds in one gpurun call)
LIB_PATH = os.environ.get("VROD_HIP_LIB") or os.path.join(_HERE, "libvrod_hip.so")

# every symbol include/vrod.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "vrod_index_create", "vrod_index_destroy", "vrod_index_reserve", "vrod_index_add",
    "vrod_index_add_synthetic", "vrod_index_count", "vrod_index_set_id_offset",
    "vrod_index_get_rows", "vrod_search", "vrod_search_device", "vrod_search_synthetic_device",
    "vrod_search_begin_device", "vrod_search_begin_synthetic_device", "vrod_search_end", "vrod_search_pending",
    "vrod_merge_topk_device", "vrod_merge_topk_packed_device", "vrod_index_set_path", "vrod_index_set_profiling",
    "vrod_index_last_stats", "vrod_index_shard_stats", "vrod_last_error", "vrod_version", "vrod_synth_rows_device",
]


class VrodError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vrod status {code}: {msg}")
        self.code = code


class SearchStats(C.Structure):
    _fields_ = [
        ("path", C.c_uint32), ("nq", C.c_uint32), ("k", C.c_uint32), ("kprime", C.c_uint32),
        ("scan_launches", C.c_uint32), ("fallback_queries", C.c_uint32),
        ("scan_ms", C.c_float), ("total_ms", C.c_float),
        ("scan_bytes", C.c_double), ("scan_flops", C.c_double),
        ("max_fast_err", C.c_float), ("eps_bound", C.c_float),
        ("split_pass", C.c_uint32), ("band_queries", C.c_uint32), ("sample_ms", C.c_float), ("exchange", C.c_uint32),
        ("overlap_ms", C.c_float),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.  If torch is
    # going to be used in this process (tests, bench) it must load first so that this library
    # binds to the same runtime; loaded the other way round the second runtime sees no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C vrod_amd/csrc). vrod_amd has no fallback path.")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.vrod_index_create.argtypes = [C.POINTER(vp), u32, i32, i32, C.POINTER(i32), i32]
    L.vrod_index_destroy.argtypes = [vp]
    L.vrod_index_reserve.argtypes = [vp, u64]
    L.vrod_index_add.argtypes = [vp, vp, u64]
    L.vrod_index_add_synthetic.argtypes = [vp, u64, u64, u64]
    L.vrod_index_count.argtypes = [vp, C.POINTER(u64)]
    L.vrod_index_set_id_offset.argtypes = [vp, u64]
    L.vrod_index_get_rows.argtypes = [vp, u64, u64, vp]
    L.vrod_search.argtypes = [vp, vp, u32, u32, vp, vp]
    L.vrod_search_device.argtypes = [vp, vp, u32, u32, vp, vp, vp]
    L.vrod_search_synthetic_device.argtypes = [vp, u64, u64, u32, u32, vp, vp, vp]
    L.vrod_search_begin_device.argtypes = [vp, vp, u32, u32, vp, vp, vp]
    L.vrod_search_begin_synthetic_device.argtypes = [vp, u64, u64, u32, u32, vp, vp, vp]
    L.vrod_search_end.argtypes = [vp]
    L.vrod_search_pending.argtypes = [vp, C.POINTER(u32)]
    L.vrod_merge_topk_device.argtypes = [i32, i32, vp, vp, u32, u32, u32, vp, vp, vp]
    L.vrod_merge_topk_packed_device.argtypes = [i32, i32, vp, u32, u32, u32, vp, vp, vp]
    L.vrod_index_set_path.argtypes = [vp, i32]
    L.vrod_index_set_profiling.argtypes = [vp, i32]
    L.vrod_index_last_stats.argtypes = [vp, C.POINTER(SearchStats)]
    L.vrod_index_shard_stats.argtypes = [vp, u32, C.POINTER(i32), C.POINTER(SearchStats)]
    L.vrod_synth_rows_device.argtypes = [i32, u64, u64, u64, u32, vp, vp]
    for name in SYMBOLS:
        getattr(L, name).restype = i32
    L.vrod_last_error.restype = C.c_char_p
    L.vrod_last_error.argtypes = []
    L.vrod_version.restype = C.c_char_p
    L.vrod_version.argtypes = []
    _lib = L
    return L


def version() -> str:
    """vrod_version(): library version + the hipcc that built the device code."""
    return load().vrod_version().decode()


def check(rc: int) -> None:
    if rc != 0:
        raise VrodError(rc, load().vrod_last_error().decode("utf-8", "replace"))
