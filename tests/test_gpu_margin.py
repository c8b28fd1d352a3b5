"""GPU test of the self-tuning candidate margin of the batched scan (vrod_index.hip: kp_boost).

The certificate needs the k-th canonical score clear of the k'-th fast score by the error bound.  At d = 3072 the bound
(4 d 2^-24 = 7.3e-4) is wider than the 12 ranks of margin k = 100 starts with (~5e-5 per rank among 100 K random unit
vectors): most queries of the first batch fail their certificate and are resolved by the band pass -- exact, but one more
scan of the corpus.  The handle doubles the margin after such a search; two searches later no certificate fails.  Results
are the oracle's bits throughout.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    import vrod_amd
    vrod_amd.load()
    return vrod_amd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_margin_follows_the_certificates(va, oracle):
    dim, n, nq, k = 3072, 100_000, 64, 100
    raw = oracle.synth_rows(1, 0, n, dim, threads=8)
    seen = []
    with va.Index(dim, "bf16", "cosine") as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        for s in range(4):
            rq = oracle.synth_rows(2, s * nq, nq, dim)
            ids, sc = ix.search(rq, k)
            st = ix.last_stats()
            seen.append((st["kprime"], st["fallback_queries"], st["band_queries"]))
            if s in (0, 3):
                oi, osc = oracle.search(raw, rq, k, 1, 0, threads=8)
                assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc)), s
    kps = [x[0] for x in seen]
    assert seen[0][0] == k + 12 and seen[0][1] > 0, seen           # the starting margin is too small here ...
    assert seen[0][1] == seen[0][2], seen                           # ... and the band pass resolved every failure
    assert kps[1] == k + 24 and kps[2] >= kps[1], seen              # the margin doubled
    assert seen[3][1] == 0, seen                                    # and no certificate fails any more


def test_widest_k_after_the_margin_was_raised(va, oracle):
    """k = VROD_MAX_K on a handle whose margin multiplier is at its largest: k' must stay inside the select windows
    (k' <= 4096), and the result exact.  The corpus is 50 copies of 1000 rows: more copies than any margin of a k = 10
    search holds (18, 26, 42), so every certificate fails (ties across the k-th boundary) and the band pass answers."""
    dim, base_n, copies = 64, 1000, 50
    base = oracle.synth_rows(5, 0, base_n, dim)
    raw = np.tile(base, (copies, 1))
    with va.Index(dim, "bf16", "cosine") as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        kps = []
        for s in range(2):                              # two searches whose certificates all fail: x2, x4
            ids, sc = ix.search(base[s * 16:(s + 1) * 16], 10)
            st = ix.last_stats()
            kps.append(st["kprime"])
            assert st["fallback_queries"] == 16
        rq = oracle.synth_rows(2, 0, 6, dim)
        k = va.MAX_K
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
        assert kps == [18, 26] and st["kprime"] == 4096, (kps, st)
        oi, osc = oracle.search(raw, rq, k, 1, 0, threads=8)
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))
