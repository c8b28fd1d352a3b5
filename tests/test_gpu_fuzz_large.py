"""GPU test: randomised searches on a corpus large enough for the staged batched scan with work stealing (1M rows), with rows
duplicated 30-fold (failed certificates, band pass, the self-tuning candidate margin all occur), k from 1 to 500, batches of
5 to 1024 queries, synchronous and pipelined: every result must be the oracle's bits.  (scripts/probes/big_fuzz.py is the long
form of this.)"""
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


@pytest.mark.parametrize("dtype,metric,seed", [("bf16", "cosine", 7), ("f32", "l2", 8)])
def test_random_searches_on_a_large_corpus_with_duplicates(va, oracle, dtype, metric, seed):
    import torch
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    dim, n, ndup = 192, 1_000_000, 2000
    DT, ME = (0 if dtype == "f32" else 1), (0 if metric == "cosine" else 1)
    raw = oracle.synth_rows(100 + seed, 0, n, dim, threads=16)
    src = rng.integers(0, n, ndup)
    for _ in range(30):
        raw[rng.integers(0, n, ndup)] = raw[src]
    prepared = oracle.prepare(raw, DT, ME, threads=16)
    seen_band = seen_boost = False
    with va.Index(dim, dtype, metric) as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        for step, (k, nq, hot, pipelined) in enumerate([(10, 1024, True, False), (1, 300, True, True), (100, 64, False, False),
                                                       (500, 1024, False, True), (10, 1024, True, True), (10, 5, True, False)]):
            rq = raw[src[rng.integers(0, ndup, nq)]] if hot else oracle.synth_rows(200 + seed, step * 2048, nq, dim)
            if pipelined:
                dq = torch.from_numpy(rq).to(dev)
                oi_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
                os_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
                ix.search_begin_device(dq, k, oi_t, os_t)
                ix.search_end()
                ids, sc = oi_t.cpu().numpy().view(np.uint64), os_t.cpu().numpy()
            else:
                ids, sc = ix.search(rq, k)
            st = ix.last_stats()
            seen_band |= st["band_queries"] > 0
            seen_boost |= st["kprime"] > k + max(16, k // 8) and not st["split_pass"]
            oi, osc = oracle.scan_topk(prepared, oracle.prepare(rq, DT, ME), k, ME, threads=16)
            assert np.array_equal(ids, oi), (step, st)
            assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32)), (step, st)
            assert st["max_fast_err"] <= st["eps_bound"], st
    assert seen_band          # the duplicates did make certificates fail ...
    assert seen_boost or dtype == "f32"   # ... and (no split pass: bf16 rows) the margin followed them
