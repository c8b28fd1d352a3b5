import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


import contextlib


@contextlib.contextmanager
def f32_split(mode):
    """VROD_F32_SPLIT for the handles created inside: "0" = fp32 MFMA pass, "1" = bf16 split
    pass forced, None = the library's default (split while memory allows)."""
    old = os.environ.get("VROD_F32_SPLIT")
    if mode is None:
        os.environ.pop("VROD_F32_SPLIT", None)
    else:
        os.environ["VROD_F32_SPLIT"] = mode
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("VROD_F32_SPLIT", None)
        else:
            os.environ["VROD_F32_SPLIT"] = old


def chunked_oracle_topk(oracle, seed, n_rows, dim, raw_queries, k, dtype, metric, chunk=1_000_000, threads=None):
    """The CPU oracle over a corpus too large to hold as fp32 on the host: the synthetic stream `seed` is generated,
    prepared and scanned one chunk of rows at a time (ids offset by the chunk's first row) and the per-chunk top-k are
    merged by the oracle's own merge -- top-k is decomposable, ties break by the smaller global id either way.
    dtype / metric: the oracle's codes (0 = f32 / cosine, 1 = bf16 / l2).  ~1 s of host time per million 768-d rows."""
    import os
    import numpy as np
    threads = threads or max(1, min(16, len(os.sched_getaffinity(0))))
    pq = oracle.prepare(raw_queries, dtype, metric)
    part_i, part_s = [], []
    for lo in range(0, n_rows, chunk):
        m = min(chunk, n_rows - lo)
        c = oracle.prepare(oracle.synth_rows(seed, lo, m, dim, threads=threads), dtype, metric, threads=threads)
        i, sc = oracle.scan_topk(c, pq, k, metric, id_offset=lo, threads=threads)
        part_i.append(i)
        part_s.append(sc)
        del c
    return oracle.merge_topk(np.stack(part_i), np.stack(part_s), metric)
