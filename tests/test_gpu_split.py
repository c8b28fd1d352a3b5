"""Opt-in split pass for fp32 corpora (VROD_F32_SPLIT=1 at handle creation): the batched fast
pass runs on the bf16 matrix cores over [hi | lo] planes of the fp32 rows (q.x ~ hi.hi + hi.lo +
lo.hi).  Only the fast pass changes: candidates are re-scored canonically and certified against a
bound that covers the representation error, so ids and score bits must still be the oracle's.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ME = {"cosine": 0, "l2": 1}


@pytest.fixture(scope="module")
def va():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    import vrod_amd
    vrod_amd.load()
    return vrod_amd


@pytest.fixture()
def split_env():
    old = os.environ.get("VROD_F32_SPLIT")
    os.environ["VROD_F32_SPLIT"] = "1"     # read when a handle is created
    yield
    if old is None:
        del os.environ["VROD_F32_SPLIT"]
    else:
        os.environ["VROD_F32_SPLIT"] = old


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", ["cosine", "l2"])
@pytest.mark.parametrize("dim,nq,k", [(100, 40, 10), (129, 300, 100), (768, 64, 10), (64, 13, 1000)])
def test_split_pass_is_bit_exact(va, oracle, split_env, metric, dim, nq, k):
    rng = np.random.default_rng(dim + nq)
    n = 30011
    raw = (rng.standard_normal((n, dim)) * rng.uniform(0.2, 3.0, (n, 1))).astype(np.float32)
    rq = rng.standard_normal((nq, dim)).astype(np.float32)
    with va.Index(dim, "f32", metric) as ix:
        ix.add(raw[:20000])
        ids0, sc0 = ix.search(rq, k)              # planes built for 20000 rows
        ix.add(raw[20000:])                       # ... and extended lazily at the next batched search
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    assert st["path"] == 2, "AUTO must route a batch of >= 13 queries to the MFMA (split) pass"
    assert st["kprime"] == min(n, k + max(32, k // 2))
    oi, osc = oracle.search(raw, rq, k, 0, ME[metric])
    assert np.array_equal(ids, oi), np.argwhere(ids != oi)[:5]
    assert np.array_equal(bits(sc), bits(osc))
    o0, s0 = oracle.search(raw[:20000], rq, k, 0, ME[metric])
    assert np.array_equal(ids0, o0) and np.array_equal(bits(sc0), bits(s0))
    assert st["max_fast_err"] <= st["eps_bound"], st
    if metric == "cosine":                          # (L2 over rows of very different norms: the bound scales with
        assert st["fallback_queries"] <= nq // 10, st   #  the largest norm and many queries take the exact path)


def test_split_is_off_by_default_and_for_small_batches(va, oracle, split_env):
    raw = oracle.synth_rows(1, 0, 20000, 96)
    rq = oracle.synth_rows(2, 0, 8, 96)
    with va.Index(96, "f32", "cosine") as ix:
        ix.add(raw)
        ix.search(rq, 10)
        assert ix.last_stats()["path"] == 1          # 8 queries: the stream scan, split or not
    os.environ["VROD_F32_SPLIT"] = "0"
    rq = oracle.synth_rows(2, 0, 40, 96)
    with va.Index(96, "f32", "cosine") as ix:
        ix.add(raw)
        ids, sc = ix.search(rq, 10)
        st = ix.last_stats()
    assert st["path"] == 2 and st["kprime"] == 10 + 16   # the fp32 MFMA pass and its k'
    oi, osc = oracle.search(raw, rq, 10, 0, 0)
    assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))
