"""ctypes binding of the CPU oracle (oracle/vrod_oracle.c) + an independent numpy check.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (vrod_amd/) never imports this.

PARITY UNPINNED BY THE REFERENCE (vRod has no scan, src/command/types.rs:127-132);
the oracle is pinned by `numpy_reference_*` below (fp64, different code path) and by
tests/golden/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvrod_oracle.so")

DTYPE_F32, DTYPE_BF16 = 0, 1
METRIC_COSINE, METRIC_L2 = 0, 1
ID_NONE = np.uint64(0xFFFFFFFFFFFFFFFF)

_lib = None


def build() -> str:
    """Compile the oracle with the committed Makefile (gcc, no FMA, no fast-math)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u64, u32, i32, f32p, u64p = C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)
        L.orc_splitmix64.restype = u64
        L.orc_splitmix64.argtypes = [u64]
        L.orc_synth_int.restype = C.c_int32
        L.orc_synth_int.argtypes = [u64, u64, u32, u32]
        L.orc_synth_rows_f32.restype = None
        L.orc_synth_rows_f32.argtypes = [u64, u64, u64, u32, f32p, i32]
        L.orc_f32_to_bf16.restype = C.c_uint16
        L.orc_f32_to_bf16.argtypes = [C.c_float]
        L.orc_prepare_rows.restype = None
        L.orc_prepare_rows.argtypes = [f32p, u64, u32, i32, i32, f32p, i32]
        L.orc_dot_canonical.restype = C.c_float
        L.orc_dot_canonical.argtypes = [f32p, f32p, u32]
        L.orc_l2_canonical.restype = C.c_float
        L.orc_l2_canonical.argtypes = [f32p, f32p, u32]
        L.orc_scan_topk.restype = i32
        L.orc_scan_topk.argtypes = [f32p, u64, u32, f32p, u32, u32, i32, u64, u64p, f32p, i32]
        L.orc_merge_topk.restype = i32
        L.orc_merge_topk.argtypes = [u64p, f32p, u32, u32, u32, i32, u64p, f32p]
        _lib = L
    return _lib


def _f32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def synth_rows(seed: int, first_row: int, n: int, dim: int, threads: int = 1) -> np.ndarray:
    """Rows [first_row, first_row+n) of synthetic stream `seed`: unit-norm fp32."""
    out = np.empty((n, dim), dtype=np.float32)
    if n:
        lib().orc_synth_rows_f32(seed, first_row, n, dim, _f32p(out), threads)
    return out


def prepare(rows: np.ndarray, dtype: int, metric: int, threads: int = 1) -> np.ndarray:
    """What insert/query do to raw fp32 vectors (normalise for cosine, bf16 round)."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.empty_like(rows)
    if rows.size:
        lib().orc_prepare_rows(_f32p(rows), rows.shape[0], rows.shape[1], dtype, metric, _f32p(out), threads)
    return out


def scan_topk(corpus: np.ndarray, queries: np.ndarray, k: int, metric: int,
              id_offset: int = 0, threads: int = 1):
    """corpus/queries are PREPARED fp32 arrays. Returns (ids u64 [nq,k], scores f32 [nq,k])."""
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    n, dim = corpus.shape if corpus.ndim == 2 else (0, queries.shape[1])
    nq = queries.shape[0]
    ids = np.empty((nq, k), dtype=np.uint64)
    sc = np.empty((nq, k), dtype=np.float32)
    rc = lib().orc_scan_topk(_f32p(corpus), n, dim, _f32p(queries), nq, k, metric, id_offset,
                             _u64p(ids), _f32p(sc), threads)
    if rc != 0:
        raise RuntimeError(f"orc_scan_topk rc={rc}")
    return ids, sc


def search(raw_corpus: np.ndarray, raw_queries: np.ndarray, k: int, dtype: int, metric: int,
           id_offset: int = 0, threads: int = 1):
    """End to end: prepare both sides, then scan."""
    return scan_topk(prepare(raw_corpus, dtype, metric, threads), prepare(raw_queries, dtype, metric, threads),
                     k, metric, id_offset, threads)


def merge_topk(ids: np.ndarray, scores: np.ndarray, metric: int):
    """ids/scores: [n_lists, nq, k] per-shard results -> merged [nq, k]."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n_lists, nq, k = ids.shape
    oi = np.empty((nq, k), dtype=np.uint64)
    os_ = np.empty((nq, k), dtype=np.float32)
    lib().orc_merge_topk(_u64p(ids), _f32p(scores), n_lists, nq, k, metric, _u64p(oi), _f32p(os_))
    return oi, os_


# ----------------------------------------------------------------------------------
# Independent restatements in numpy (different code path; used to pin the C oracle).
# ----------------------------------------------------------------------------------

def numpy_splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def numpy_synth_rows(seed: int, first_row: int, n: int, dim: int) -> np.ndarray:
    key = numpy_splitmix64(np.array([seed], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        idx = (np.arange(first_row, first_row + n, dtype=np.uint64)[:, None] * np.uint64(dim)
               + np.arange(dim, dtype=np.uint64)[None, :])
    h = numpy_splitmix64(key ^ idx)
    m = np.uint64(0xFFFF)
    s = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48))).astype(np.int64) - 131070
    v = s.astype(np.float64)
    ss = (s * s).sum(axis=1).astype(np.float64)  # exact integers
    nrm = np.sqrt(ss)
    nrm[nrm == 0] = 1.0
    return (v / nrm[:, None]).astype(np.float32)


def numpy_bf16_round(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)
    return (r.astype(np.uint32) << np.uint32(16)).view(np.float32)


def numpy_prepare(rows: np.ndarray, dtype: int, metric: int) -> np.ndarray:
    rows = np.asarray(rows, dtype=np.float32)
    out = rows
    if metric == METRIC_COSINE:
        x = rows.astype(np.float64)
        ss = np.zeros(rows.shape[0], dtype=np.float64)
        for j in range(rows.shape[1]):  # left to right, like the spec
            ss = ss + x[:, j] * x[:, j]
        nrm = np.sqrt(ss)
        safe = np.where(nrm == 0, 1.0, nrm)
        out = np.where((nrm == 0)[:, None], 0.0, x / safe[:, None]).astype(np.float32)
    if dtype == DTYPE_BF16:
        out = numpy_bf16_round(out)
    return out


def numpy_scores_canonical(corpus: np.ndarray, queries: np.ndarray, metric: int) -> np.ndarray:
    """fp32 scores in the canonical order, vectorised across rows (sequential over j)."""
    corpus = np.asarray(corpus, dtype=np.float32)
    queries = np.asarray(queries, dtype=np.float32)
    out = np.zeros((queries.shape[0], corpus.shape[0]), dtype=np.float32)
    for j in range(corpus.shape[1]):
        if metric == METRIC_COSINE:
            p = (queries[:, j:j + 1] * corpus[None, :, j]).astype(np.float32)
        else:
            df = (queries[:, j:j + 1] - corpus[None, :, j]).astype(np.float32)
            p = (df * df).astype(np.float32)
        out = (out + p).astype(np.float32)
    return out


def numpy_topk_from_scores(scores: np.ndarray, k: int, metric: int, id_offset: int = 0):
    """Exact top-k with the spec's ordering (best first, ties -> smaller id)."""
    nq, n = scores.shape
    ids = np.full((nq, k), ID_NONE, dtype=np.uint64)
    sc = np.full((nq, k), np.nan, dtype=np.float32)
    for qi in range(nq):
        s = scores[qi]
        key = -s.astype(np.float64) if metric == METRIC_COSINE else s.astype(np.float64)
        order = np.lexsort((np.arange(n), key))  # primary key, then id
        m = min(k, n)
        ids[qi, :m] = order[:m].astype(np.uint64) + np.uint64(id_offset)
        sc[qi, :m] = s[order[:m]]
    return ids, sc


def numpy_scores_fp64(corpus: np.ndarray, queries: np.ndarray, metric: int) -> np.ndarray:
    c = np.asarray(corpus, dtype=np.float64)
    q = np.asarray(queries, dtype=np.float64)
    if metric == METRIC_COSINE:
        return q @ c.T
    return ((q * q).sum(1)[:, None] + (c * c).sum(1)[None, :] - 2.0 * (q @ c.T))
