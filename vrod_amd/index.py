"""Python harness over the C ABI (include/vrod.h): device memory plumbing only.

`Index` owns one `vrod_index*` (one GPU).  numpy in / numpy out goes through
`vrod_search`; torch device tensors go through `vrod_search_device` (torch is plumbing
here: device buffers, streams and torch.distributed for the RCCL all-gather).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import SearchStats, VrodError, check

DTYPE_F32, DTYPE_BF16 = 0, 1
METRIC_COSINE, METRIC_L2 = 0, 1
PATH_AUTO, PATH_STREAM, PATH_MFMA, PATH_EXACT = 0, 1, 2, 3
ID_NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
MAX_K = 3584

_DTYPES = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "float32": DTYPE_F32, "bf16": DTYPE_BF16, "bfloat16": DTYPE_BF16}
_METRICS = {"cosine": METRIC_COSINE, "cos": METRIC_COSINE, "l2": METRIC_L2, "euclidean": METRIC_L2}


def _enum(v, table, what):
    if isinstance(v, str):
        try:
            return table[v.lower()]
        except KeyError:
            raise ValueError(f"unknown {what} {v!r}") from None
    return int(v)


class Index:
    """One shard of a brute-force index on one MI355X."""

    def __init__(self, dim: int, dtype="f32", metric="cosine", device: int = 0, devices=None):
        """`devices` (a list of device ids, repeats allowed) makes ONE handle that deals its rows to
        several GPUs and searches them all per call (vrod_index_create with n_devices > 1)."""
        self._L = _lib.load()
        self._h = C.c_void_p()
        self.dim = int(dim)
        self.dtype = _enum(dtype, _DTYPES, "dtype")
        self.metric = _enum(metric, _METRICS, "metric")
        ids = [int(d) for d in devices] if devices is not None else [int(device)]
        self.device = ids[0]
        dev = (C.c_int * len(ids))(*ids)
        check(self._L.vrod_index_create(C.byref(self._h), self.dim, self.dtype, self.metric, dev, len(ids)))

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.vrod_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- corpus
    def reserve(self, n: int):
        check(self._L.vrod_index_reserve(self._h, int(n)))

    def add(self, rows: np.ndarray):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}]")
        check(self._L.vrod_index_add(self._h, rows.ctypes.data_as(C.c_void_p), rows.shape[0]))

    def add_synthetic(self, seed: int, first_row: int, n: int):
        check(self._L.vrod_index_add_synthetic(self._h, int(seed), int(first_row), int(n)))

    @property
    def count(self) -> int:
        out = C.c_uint64()
        check(self._L.vrod_index_count(self._h, C.byref(out)))
        return out.value

    def set_id_offset(self, off: int):
        check(self._L.vrod_index_set_id_offset(self._h, int(off)))

    def get_rows(self, first: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), dtype=np.float32)
        check(self._L.vrod_index_get_rows(self._h, int(first), int(n), out.ctypes.data_as(C.c_void_p)))
        return out

    # -- knobs
    def set_path(self, path: int):
        check(self._L.vrod_index_set_path(self._h, int(path)))

    def set_profiling(self, level):
        """0/False: off; 1/True: scan_ms (events attached to the scan dispatches); 2: + total_ms."""
        check(self._L.vrod_index_set_profiling(self._h, int(level)))

    def last_stats(self) -> dict:
        st = SearchStats()
        check(self._L.vrod_index_last_stats(self._h, C.byref(st)))
        return st.as_dict()

    def shard_stats(self, shard: int) -> dict:
        """Counters of one shard of the last completed search + the device it lives on."""
        st = SearchStats()
        dev = C.c_int32(-1)
        check(self._L.vrod_index_shard_stats(self._h, int(shard), C.byref(dev), C.byref(st)))
        d = st.as_dict()
        d["device"] = dev.value
        return d

    # -- search
    def search(self, queries: np.ndarray, k: int):
        """numpy [nq, dim] fp32 -> (ids uint64 [nq, k], scores float32 [nq, k])."""
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim == 1:
            queries = queries[None, :]
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"queries must be [nq, {self.dim}]")
        nq = queries.shape[0]
        ids = np.empty((nq, k), dtype=np.uint64)
        sc = np.empty((nq, k), dtype=np.float32)
        check(self._L.vrod_search(self._h, queries.ctypes.data_as(C.c_void_p), nq, int(k),
                                  ids.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)))
        return ids, sc

    def search_device(self, d_queries, k: int, out_ids=None, out_scores=None):
        """torch CUDA tensor [nq, dim] fp32 -> (ids int64-viewed-uint64 [nq,k], scores [nq,k]) on device."""
        import torch
        assert d_queries.is_cuda and d_queries.dtype == torch.float32 and d_queries.is_contiguous()
        nq = d_queries.shape[0]
        if out_ids is None:
            out_ids = torch.empty((nq, k), dtype=torch.int64, device=d_queries.device)
        if out_scores is None:
            out_scores = torch.empty((nq, k), dtype=torch.float32, device=d_queries.device)
        stream = torch.cuda.current_stream(d_queries.device).cuda_stream
        check(self._L.vrod_search_device(self._h, d_queries.data_ptr(), nq, int(k), out_ids.data_ptr(),
                                         out_scores.data_ptr(), C.c_void_p(stream)))
        return out_ids, out_scores

    def search_synthetic_device(self, seed: int, first_row: int, nq: int, k: int, out_ids, out_scores):
        """Queries = rows [first_row, first_row+nq) of synthetic stream `seed`, generated on device."""
        import torch
        stream = torch.cuda.current_stream(out_ids.device).cuda_stream
        check(self._L.vrod_search_synthetic_device(self._h, int(seed), int(first_row), int(nq), int(k),
                                                   out_ids.data_ptr(), out_scores.data_ptr(), C.c_void_p(stream)))
        return out_ids, out_scores

    # -- pipelined search: begin(s+1) before end(s) keeps the device busy between batches
    def search_begin_device(self, d_queries, k: int, out_ids, out_scores):
        import torch
        assert d_queries.is_cuda and d_queries.dtype == torch.float32 and d_queries.is_contiguous()
        stream = torch.cuda.current_stream(d_queries.device).cuda_stream
        check(self._L.vrod_search_begin_device(self._h, d_queries.data_ptr(), d_queries.shape[0], int(k),
                                               out_ids.data_ptr(), out_scores.data_ptr(), C.c_void_p(stream)))

    def search_begin_synthetic_device(self, seed: int, first_row: int, nq: int, k: int, out_ids, out_scores):
        import torch
        stream = torch.cuda.current_stream(out_ids.device).cuda_stream
        check(self._L.vrod_search_begin_synthetic_device(self._h, int(seed), int(first_row), int(nq), int(k),
                                                         out_ids.data_ptr(), out_scores.data_ptr(), C.c_void_p(stream)))

    def search_end(self):
        """Complete the oldest pending search; its output tensors are final when this returns."""
        check(self._L.vrod_search_end(self._h))

    @property
    def pending(self) -> int:
        out = C.c_uint32()
        check(self._L.vrod_search_pending(self._h, C.byref(out)))
        return out.value


def merge_topk_device(device: int, metric, ids, scores, out_ids=None, out_scores=None):
    """ids/scores: torch CUDA tensors [n_lists, nq, k] (int64 bits of uint64 / float32) -> merged [nq, k]."""
    import torch
    L = _lib.load()
    n_lists, nq, k = ids.shape
    assert ids.is_contiguous() and scores.is_contiguous()
    if out_ids is None:
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=ids.device)
    if out_scores is None:
        out_scores = torch.empty((nq, k), dtype=torch.float32, device=ids.device)
    stream = torch.cuda.current_stream(ids.device).cuda_stream
    check(L.vrod_merge_topk_device(int(device), _enum(metric, _METRICS, "metric"), ids.data_ptr(), scores.data_ptr(),
                                   n_lists, nq, k, out_ids.data_ptr(), out_scores.data_ptr(), C.c_void_p(stream)))
    return out_ids, out_scores


def merge_topk_packed_device(device: int, metric, packed, n_lists: int, nq: int, k: int, out_ids, out_scores):
    """packed: torch CUDA uint8 buffer of n_lists blocks (shard.alloc_packed layout) -> merged [nq, k]."""
    import torch
    L = _lib.load()
    assert packed.is_contiguous() and packed.numel() == n_lists * 12 * nq * k
    stream = torch.cuda.current_stream(packed.device).cuda_stream
    check(L.vrod_merge_topk_packed_device(int(device), _enum(metric, _METRICS, "metric"), packed.data_ptr(), n_lists, nq, k,
                                          out_ids.data_ptr(), out_scores.data_ptr(), C.c_void_p(stream)))
    return out_ids, out_scores


def synth_rows_device(device: int, seed: int, first_row: int, n: int, dim: int):
    """Synthetic rows generated on the device, returned as a torch tensor [n, dim]."""
    import torch
    L = _lib.load()
    out = torch.empty((n, dim), dtype=torch.float32, device=f"cuda:{device}")
    stream = torch.cuda.current_stream(out.device).cuda_stream
    check(L.vrod_synth_rows_device(int(device), int(seed), int(first_row), int(n), int(dim), out.data_ptr(),
                                   C.c_void_p(stream)))
    return out
