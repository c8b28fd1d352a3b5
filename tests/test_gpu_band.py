"""The band pass: second chance of queries whose certificate failed (exact duplicates / near-ties around the
k-th result).  One more filtered scan shared by all failed queries collects the rows within the error bound of
the k-th canonical score; their canonical re-score + exact select is the answer.  Results must stay the bits of
the oracle, and most failed queries must be resolved by it (band_queries) instead of the one-pass-per-8-queries
exact path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = {"f32": 0, "bf16": 1}
ME = {"cosine": 0, "l2": 1}


@pytest.fixture(scope="module")
def va():
    import torch
    assert torch.cuda.is_available()
    import vrod_amd
    vrod_amd.load()
    return vrod_amd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def dup_corpus(oracle, dim, n_base, groups, copies, seed=21):
    """n_base distinct rows followed by `copies` copies of each of the first `groups` rows."""
    base = oracle.synth_rows(seed, 0, n_base, dim, threads=8)
    return np.concatenate([base] + [base[:groups]] * copies)


@pytest.mark.parametrize("dtype,metric,split", [("bf16", "cosine", None), ("bf16", "l2", None), ("f32", "cosine", "1"), ("f32", "l2", "0")])
def test_duplicate_heavy_batch_is_resolved_by_the_band_pass(va, oracle, dtype, metric, split):
    from conftest import f32_split
    dim, n_base, groups, copies, k = 96, 60000, 400, 60, 10
    raw = dup_corpus(oracle, dim, n_base, groups, copies)          # 84000 rows; 400 rows exist 61 times (more than the k' = 26 or 42 candidates a query keeps)
    rng = np.random.default_rng(3)
    # 200 queries sit on duplicated rows (all 61 copies tie at the top), 56 are random
    rq = np.concatenate([raw[rng.integers(0, groups, 200)], oracle.synth_rows(22, 0, 56, dim)]).astype(np.float32)
    oi, osc = oracle.search(raw, rq, k, DT[dtype], ME[metric], threads=8)
    with f32_split(split), va.Index(dim, dtype, metric) as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    assert np.array_equal(ids, oi), np.argwhere(ids != oi)[:5]
    assert np.array_equal(bits(sc), bits(osc))
    assert st["fallback_queries"] >= 200
    assert st["band_queries"] >= 200, st                            # ... and none of them needed a pass of the exact path
    # the band scan's own fast-vs-canonical differences are part of max_fast_err: the band is a superset of the true
    # top-k only while they stay inside the bound the band was cut with
    assert 0 < st["max_fast_err"] <= st["eps_bound"], st


def test_band_too_wide_stays_on_the_exact_path(va, oracle):
    """5000 copies of one row: the band of a query on it holds more rows than the re-score + select take (4096):
    that query is answered by the exact path, the others by the band pass.  On an fp32 corpus without bf16 planes
    (the scan runs at the fp32 matrix rate) a handful of failed queries skips the band pass altogether."""
    from conftest import f32_split
    dim, k = 64, 10
    base = oracle.synth_rows(31, 0, 30000, dim, threads=4)
    raw = np.concatenate([base, np.repeat(base[:1], 5000, axis=0), np.repeat(base[1:41], 40, axis=0)])
    rq = np.concatenate([base[:1], base[1:41], oracle.synth_rows(32, 0, 23, dim)]).astype(np.float32)   # 1 wide band, 40 narrow, 23 clean
    oi, osc = oracle.search(raw, rq, k, 1, 0, threads=8)
    with va.Index(dim, "bf16", "cosine") as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))
        assert st["fallback_queries"] >= 41 and 40 <= st["band_queries"] < st["fallback_queries"], st
        ids, sc = ix.search(rq[1:4], k)                              # three failed queries: still the band pass (one HBM-bound pass)
        st = ix.last_stats()
        assert np.array_equal(ids, oi[1:4]) and np.array_equal(bits(sc), bits(osc[1:4]))
        assert st["fallback_queries"] == 3 and st["band_queries"] == 3, st
    oi, osc = oracle.search(raw, rq, k, 0, 0, threads=8)
    with f32_split("0"), va.Index(dim, "f32", "cosine") as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        ids, sc = ix.search(rq[:40], k)                              # 40 failed queries on the fp32 matrix rate: exact path
        st = ix.last_stats()
    assert np.array_equal(ids, oi[:40]) and np.array_equal(bits(sc), bits(osc[:40]))
    assert st["fallback_queries"] >= 40 and st["band_queries"] == 0, st
