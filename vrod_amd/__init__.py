"""vrod_amd -- MI355X (gfx950) brute-force similarity scan + top-k behind vRod's
SEARCHSIMILAR command (sekulas/vRod src/command/types.rs:121-132).

The product is libvrod_hip.so (hand-written HIP, C ABI in include/vrod.h); this package is
the thin harness that loads it.  No CPU or PyTorch fallback exists.
"""
from ._lib import LIB_PATH, SYMBOLS, SearchStats, VrodError, load, version  # noqa: F401
from .index import (DTYPE_BF16, DTYPE_F32, ID_NONE, MAX_K, METRIC_COSINE, METRIC_L2,  # noqa: F401
                    PATH_AUTO, PATH_EXACT, PATH_MFMA, PATH_STREAM, Index, merge_topk_device,
                    merge_topk_packed_device, synth_rows_device)

__all__ = ["Index", "VrodError", "merge_topk_device", "merge_topk_packed_device", "synth_rows_device", "load", "LIB_PATH", "SYMBOLS"]
