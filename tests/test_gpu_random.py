"""Randomised parity sweep on the GPU: random shapes (including awkward ones: dim not a
multiple of the 128-B line, rows not of the tile, k up to N, batches around the 8 / 256
boundaries), dtypes, metrics and forced paths, every result compared bit for bit with the
oracle.  Seeds are fixed: failures reproduce."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = {"f32": 0, "bf16": 1}
ME = {"cosine": 0, "l2": 1}


def _cases():
    # VROD_RANDOM_SEED / VROD_RANDOM_CASES: one-off larger sweeps (the defaults are the committed suite)
    import os
    rng = np.random.default_rng(int(os.environ.get("VROD_RANDOM_SEED", "20261004")))
    out = []
    for i in range(int(os.environ.get("VROD_RANDOM_CASES", "36"))):
        n = int(rng.choice([1, 2, 63, 257, 1000, 4097, 9000, 20011, 70001]))
        dim = int(rng.choice([1, 3, 31, 32, 33, 64, 100, 129, 257, 384]))
        nq = int(rng.choice([1, 2, 7, 8, 9, 17, 255, 256, 257]))
        k = int(rng.choice([1, 2, 10, 33, 100, 400]))
        dtype = ["f32", "bf16"][int(rng.integers(2))]
        metric = ["cosine", "l2"][int(rng.integers(2))]
        path = int(rng.choice([0, 1, 2]))
        scale = float(rng.choice([1.0, 1e-3, 50.0]))
        dup = bool(rng.integers(4) == 0)
        if n * nq * dim > 2.5e9:      # keep the single-threaded oracle side in seconds
            nq = max(1, int(2.5e9 / (n * dim)))
        if path == 1 and nq > 64:     # forced stream path scans the corpus once per 8 queries
            nq = 64
        out.append((i, n, dim, nq, k, dtype, metric, path, scale, dup))
    return out


@pytest.fixture(scope="module")
def va():
    import vrod_amd
    vrod_amd.load()
    return vrod_amd


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c[0]}-n{c[1]}-d{c[2]}-q{c[3]}-k{c[4]}-{c[5]}-{c[6]}-p{c[7]}")
def test_random_case_matches_oracle(va, oracle, case):
    i, n, dim, nq, k, dtype, metric, path, scale, dup = case
    rng = np.random.default_rng(1000 + i)
    raw = (rng.standard_normal((n, dim)) * scale).astype(np.float32)
    if dup and n > 4:
        raw[n // 2:] = raw[: n - n // 2]          # duplicated rows: exact ties across the corpus
    rq = (rng.standard_normal((nq, dim)) * scale).astype(np.float32)
    if dup:
        rq[0] = raw[0]                             # a query equal to a stored (duplicated) row
    from conftest import f32_split
    with f32_split([None, "0", "1"][i % 3]), va.Index(dim, dtype, metric) as ix:   # fp32 batches: default / fp32 MFMA / forced split
        ix.add(raw)
        ix.set_path(path)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    oi, osc = oracle.search(raw, rq, k, DT[dtype], ME[metric], threads=8)
    assert np.array_equal(ids, oi), f"ids differ: {np.argwhere(ids != oi)[:4]} stats={st}"
    assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32)), f"score bits differ, stats={st}"
    if st["fallback_queries"] == 0 and st["path"] != 3:
        assert st["max_fast_err"] <= st["eps_bound"] * 1.0001, st
