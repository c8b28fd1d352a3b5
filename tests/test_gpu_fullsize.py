"""Properties that do not depend on the size, checked at BASELINE.json's FULL sizes.

The CPU oracle walks one (query, row) pair at a time: at 10M rows it is good for a handful of
queries, not for a batch of 1024.  What can be checked at full size without it:

* independent kernels agree bit for bit: the MFMA batch (fast pass + certificate), the HBM stream
  scan (another fast pass), and the EXACT path (canonical score of every row + exact select, no
  fast pass and no certificate at all) -- three different routes to the same top-k;
* decomposition: the top-k of the whole corpus is the merge of the top-k of its row shards
   -- the checksum-of-checksums of this domain;
* planted rows: a query that IS a stored row comes back first, with the self-score the oracle
  computes for that row on its own;
* order: best first, ties by ascending id, ids unique and in range; top-k is a prefix of top-k2.

cfg2 (1M x 768 fp32, one query) is small enough for the oracle itself and is compared directly.
Since round 3 the three big configs are ALSO compared with the oracle directly at full size, for a few queries each:
the oracle walks the corpus in chunks of a million rows (conftest.chunked_oracle_topk) and merges the per-chunk lists.
Corpus = synthetic stream 1, queries = stream 2, generated on the device by the
same generator the oracle holds (tests/test_gpu_parity.py::test_synth_stream_matches_oracle).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CORPUS_SEED, QUERY_SEED = 1, 2


@pytest.fixture(scope="module")
def va():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    import vrod_amd
    vrod_amd.load()
    return vrod_amd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def search_syn(ix, seed, first, nq, k, path=0):
    """Queries = rows [first, first + nq) of stream `seed`, made on the device."""
    import torch
    dev = torch.device("cuda", 0)
    oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
    osc = torch.empty((nq, k), dtype=torch.float32, device=dev)
    ix.set_path(path)
    ix.search_synthetic_device(seed, first, nq, k, oi, osc)
    st = ix.last_stats()
    ix.set_path(0)
    return oi.cpu().numpy().view(np.uint64), osc.cpu().numpy(), st


def check_order(ids, sc, n_rows, metric):
    better = (lambda a, b: a >= b) if metric == "cosine" else (lambda a, b: a <= b)
    assert ids.max() < n_rows
    assert np.all(better(sc[:, :-1], sc[:, 1:])), "scores are not best-first"
    tie = sc[:, :-1] == sc[:, 1:]
    assert np.all(ids[:, :-1][tie] < ids[:, 1:][tie]), "ties are not in ascending id order"
    srt = np.sort(ids, axis=1)
    assert np.all(srt[:, :-1] != srt[:, 1:]), "an id is reported twice"


# ------------------------------------------------------------------ cfg3: 10M x 768 bf16 cosine, batch 1024, top-10
N3, D3, Q3, K3 = 10_000_000, 768, 1024, 10


@pytest.fixture(scope="module")
def cfg3(va):
    ix = va.Index(D3, "bf16", "cosine")
    ix.add_synthetic(CORPUS_SEED, 0, N3)
    yield ix
    ix.close()


@pytest.fixture(scope="module")
def cfg3_batch(cfg3):
    ids, sc, st = search_syn(cfg3, QUERY_SEED, 0, Q3, K3)
    assert st["path"] == 2 and st["nq"] == Q3
    return ids, sc, st


def test_cfg3_order_and_certificates(cfg3_batch):
    ids, sc, st = cfg3_batch
    check_order(ids, sc, N3, "cosine")
    assert st["fallback_queries"] == 0, "random unit vectors: every certificate should hold"
    assert 0 < st["max_fast_err"] <= st["eps_bound"]


def test_cfg3_three_routes_agree(cfg3, cfg3_batch):
    ids, sc, _ = cfg3_batch
    ie, se, st = search_syn(cfg3, QUERY_SEED, 0, 24, K3, path=3)     # exact path: 3 passes of 8 queries
    assert st["path"] == 3 and st["fallback_queries"] == 24
    assert np.array_equal(ie, ids[:24]) and np.array_equal(bits(se), bits(sc[:24])), "MFMA batch != exact path"
    i1, s1, st = search_syn(cfg3, QUERY_SEED, 0, 4, K3, path=1)      # stream scan
    assert st["path"] == 1
    assert np.array_equal(i1, ids[:4]) and np.array_equal(bits(s1), bits(sc[:4])), "MFMA batch != stream scan"
    # a window of the batch further in: the same queries as their own, smaller batch
    iw, sw, _ = search_syn(cfg3, QUERY_SEED, 700, 300, K3)
    assert np.array_equal(iw, ids[700:1000]) and np.array_equal(bits(sw), bits(sc[700:1000]))


def test_cfg3_full_size_against_the_chunked_oracle(oracle, cfg3_batch):
    """ids AND score bits of 8 queries of the 1024-batch over all 10M rows = the CPU oracle (bf16 data, canonical fp32 chain)"""
    from conftest import chunked_oracle_topk
    ids, sc, _ = cfg3_batch
    qs = [0, 1, 137, 511, 700, 1000, 1022, 1023]
    rq = np.concatenate([oracle.synth_rows(QUERY_SEED, q, 1, D3) for q in qs])
    oi, osc = chunked_oracle_topk(oracle, CORPUS_SEED, N3, D3, rq, K3, 1, 0)
    assert np.array_equal(ids[qs], oi), np.argwhere(ids[qs] != oi)[:5]
    assert np.array_equal(bits(sc[qs]), bits(osc))


def test_cfg3_topk_is_a_prefix_of_top100(cfg3, cfg3_batch):
    ids, sc, _ = cfg3_batch
    i100, s100, _ = search_syn(cfg3, QUERY_SEED, 0, 64, 100)
    check_order(i100, s100, N3, "cosine")
    assert np.array_equal(i100[:, :K3], ids[:64]) and np.array_equal(bits(s100[:, :K3]), bits(sc[:64]))


def test_cfg3_planted_rows_come_back_first(va, oracle, cfg3):
    r0, m = 7_654_321, 48
    ids, sc, _ = search_syn(cfg3, CORPUS_SEED, r0, m, K3)             # queries = stored rows r0 .. r0+m-1
    assert np.array_equal(ids[:, 0], np.arange(r0, r0 + m, dtype=np.uint64))
    rows = oracle.synth_rows(CORPUS_SEED, r0, m, D3)
    _, self_sc = oracle.search(rows, rows, 1, 1, 0)                   # the oracle's score of each row with itself
    assert np.array_equal(bits(sc[:, 0]), bits(self_sc[:, 0]))
    check_order(ids, sc, N3, "cosine")


def test_cfg3_merge_of_shards_is_the_whole(va, cfg3, cfg3_batch):
    """Three unequal row shards with their id offsets -> vrod_merge_topk_device == one handle."""
    import torch
    ids, sc, _ = cfg3_batch
    cuts = [0, 2_500_000, 6_000_001, N3]
    dev = torch.device("cuda", 0)
    li = torch.empty((3, Q3, K3), dtype=torch.int64, device=dev)
    ls = torch.empty((3, Q3, K3), dtype=torch.float32, device=dev)
    for g in range(3):
        with va.Index(D3, "bf16", "cosine") as sh:
            sh.add_synthetic(CORPUS_SEED, cuts[g], cuts[g + 1] - cuts[g])
            sh.set_id_offset(cuts[g])
            sh.search_synthetic_device(QUERY_SEED, 0, Q3, K3, li[g], ls[g])
    mi, ms = va.merge_topk_device(0, "cosine", li, ls)
    assert np.array_equal(mi.cpu().numpy().view(np.uint64), ids)
    assert np.array_equal(bits(ms.cpu().numpy()), bits(sc))


# ------------------------------------------------------------------ cfg2: 1M x 768 fp32 L2, one query, top-100
def test_cfg2_full_size_against_the_oracle(va, oracle):
    n, d, k = 1_000_000, 768, 100
    raw = oracle.synth_rows(CORPUS_SEED, 0, n, d, threads=8)
    rq = oracle.synth_rows(QUERY_SEED, 0, 2, d)
    oi, osc = oracle.search(raw, rq, k, 0, 1, threads=8)
    with va.Index(d, "f32", "l2") as ix:
        ix.add_synthetic(CORPUS_SEED, 0, n)
        for path in (0, 1, 2, 3):
            ids, sc, st = search_syn(ix, QUERY_SEED, 0, 2, k, path=path)
            assert st["path"] == (1 if path == 0 else path)
            assert np.array_equal(ids, oi), f"path {path}: ids differ"
            assert np.array_equal(bits(sc), bits(osc)), f"path {path}: score bits differ"
        check_order(ids, sc, n, "l2")


# ------------------------------------------------------------------ cfg5: 10M x 1536 fp32 cosine, batch 256, top-1000
def test_cfg5_three_routes_agree(va, oracle):
    n, d, nq, k = 10_000_000, 1536, 256, 1000
    with va.Index(d, "f32", "cosine") as ix:
        ix.add_synthetic(CORPUS_SEED, 0, n)
        ids, sc, st = search_syn(ix, QUERY_SEED, 0, nq, k)
        assert st["path"] == 2 and st["split_pass"] == 1, "288 GB of HBM: the default is the bf16 split pass"
        assert st["fallback_queries"] == 0 and 0 < st["max_fast_err"] <= st["eps_bound"]
        check_order(ids, sc, n, "cosine")
        ie, se, st = search_syn(ix, QUERY_SEED, 0, 8, k, path=3)
        assert st["fallback_queries"] == 8
        assert np.array_equal(ie, ids[:8]) and np.array_equal(bits(se), bits(sc[:8])), "split pass != exact path"
        i1, s1, st = search_syn(ix, QUERY_SEED, 8, 8, k, path=1)
        assert st["path"] == 1
        assert np.array_equal(i1, ids[8:16]) and np.array_equal(bits(s1), bits(sc[8:16])), "split pass != stream scan"
    # two queries of the batch against the CPU oracle over all 10M x 1536 fp32 rows, k = 1000 (ids and score bits)
    from conftest import chunked_oracle_topk
    qs = [0, 255]
    rq = np.concatenate([oracle.synth_rows(QUERY_SEED, q, 1, d) for q in qs])
    oi, osc = chunked_oracle_topk(oracle, CORPUS_SEED, n, d, rq, k, 0, 0, chunk=500_000)
    assert np.array_equal(ids[qs], oi), np.argwhere(ids[qs] != oi)[:5]
    assert np.array_equal(bits(sc[qs]), bits(osc))
    # the same corpus without planes: the fp32 matrix-core pass
    from conftest import f32_split
    with f32_split("0"), va.Index(d, "f32", "cosine") as ix:
        ix.add_synthetic(CORPUS_SEED, 0, n)
        i2, s2, st = search_syn(ix, QUERY_SEED, 0, nq, k)
        assert st["path"] == 2 and st["split_pass"] == 0 and st["fallback_queries"] == 0
    assert np.array_equal(i2, ids) and np.array_equal(bits(s2), bits(sc)), "split pass != fp32 MFMA pass"


# ------------------------------------------------------------------ cfg4: 40M x 768 bf16 cosine, batch 1024, 8 shards of 5M
def test_cfg4_eight_shards_merge_to_one_handle(va, oracle):
    """cfg4's own decomposition, on one device: the 8 x 5M-row shards (searched one after the
    other, ids offset by 5M each) merged by vrod_merge_topk_packed_device -- the layout the RCCL
    all-gather delivers -- against ONE handle holding all 40M rows (61 GB resident)."""
    import torch
    from vrod_amd.shard import alloc_packed, shard_range
    n, d, nq, k, world = 40_000_000, 768, 1024, 10, 8
    dev = torch.device("cuda", 0)
    gathered = torch.empty(world * 12 * nq * k, dtype=torch.uint8, device=dev)
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        packed, oi, osc = alloc_packed(nq, k, dev)
        with va.Index(d, "bf16", "cosine") as sh:
            sh.add_synthetic(CORPUS_SEED, lo, hi - lo)
            sh.set_id_offset(lo)
            sh.search_synthetic_device(QUERY_SEED, 0, nq, k, oi, osc)
        gathered[r * packed.numel():(r + 1) * packed.numel()] = packed
    mi = torch.empty((nq, k), dtype=torch.int64, device=dev)
    ms = torch.empty((nq, k), dtype=torch.float32, device=dev)
    va.merge_topk_packed_device(0, "cosine", gathered, world, nq, k, mi, ms)
    with va.Index(d, "bf16", "cosine") as ix:
        ix.add_synthetic(CORPUS_SEED, 0, n)
        ids, sc, st = search_syn(ix, QUERY_SEED, 0, nq, k)
    assert st["fallback_queries"] == 0
    check_order(ids, sc, n, "cosine")
    assert np.array_equal(mi.cpu().numpy().view(np.uint64), ids)
    assert np.array_equal(bits(ms.cpu().numpy()), bits(sc))
    # two queries of the batch against the CPU oracle over all 40M rows
    from conftest import chunked_oracle_topk
    qs = [0, 1023]
    rq = np.concatenate([oracle.synth_rows(QUERY_SEED, q, 1, d) for q in qs])
    oi, osc = chunked_oracle_topk(oracle, CORPUS_SEED, n, d, rq, k, 1, 0, chunk=2_000_000)
    assert np.array_equal(ids[qs], oi), np.argwhere(ids[qs] != oi)[:5]
    assert np.array_equal(bits(sc[qs]), bits(osc))


def test_cfg3_with_duplicates_full_size_band_pass_properties(va, oracle):
    """10M x 768 bf16 cosine with 5 % duplicates (64 extra copies of rows 0..7811), a batch of 1024 queries that
    ARE duplicated rows: all 65 copies tie at the top, no certificate holds, the whole batch takes the band pass.
    Checked without the oracle: every query's top-10 are the ten SMALLEST ids among its 65 copies (the copies
    of row r sit at r and at base + c * groups + r), all with the score of the row against itself, which the
    oracle computes for that one row; and a sample of the queries agrees bit for bit with the EXACT path
    (canonical score of every row, no fast pass, no certificate, no band)."""
    n, dim, k, nq, groups, copies = 10_000_000, 768, 10, 1024, 7812, 64
    base = n - groups * copies
    with va.Index(dim, "bf16", "cosine") as ix:
        ix.reserve(n)
        ix.add_synthetic(CORPUS_SEED, 0, base)
        for _ in range(copies):
            ix.add_synthetic(CORPUS_SEED, 0, groups)
        assert ix.count == n
        first = 3000
        ids, sc, st = search_syn(ix, CORPUS_SEED, first, nq, k, path=va.PATH_MFMA)
        assert st["fallback_queries"] == nq and st["band_queries"] == nq, st     # nobody needed a pass of the exact path
        r = np.arange(first, first + nq, dtype=np.uint64)
        want = np.concatenate([r[:, None], base + np.arange(k - 1, dtype=np.uint64)[None, :] * groups + r[:, None]], axis=1)
        assert np.array_equal(ids, want)
        rows = oracle.prepare(oracle.synth_rows(CORPUS_SEED, first, 4, dim), 1, 0)
        self_score = oracle.scan_topk(rows[:1], rows[:1], 1, 0)[1][0, 0]
        assert np.all(bits(sc[0]) == bits(np.float32(self_score)))
        assert np.all(sc[:, :1] == sc), "the copies of a row do not all carry the same score"
        check_order(ids, sc, n, "cosine")
        ei, es, est = search_syn(ix, CORPUS_SEED, first, 8, k, path=va.PATH_EXACT)
        assert np.array_equal(ei, ids[:8]) and np.array_equal(bits(es), bits(sc[:8]))
