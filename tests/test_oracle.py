"""CPU tests of the oracle (the parity anchor): golden fixtures, an independent numpy
restatement, and an fp64 scikit-learn brute force as the outside cross-check.

vRod has no tests or golden vectors for this path: PARITY UNPINNED by the
reference; these are what pins the oracle instead.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden.npz"), allow_pickle=False)
META = json.load(open(os.path.join(HERE, "golden", "golden_meta.json")))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", sorted(META))
def test_oracle_reproduces_golden(oracle, name):
    m = META[name]
    if m.get("raw"):
        raw, rq = GOLD[name + "__raw"], GOLD[name + "__queries"]
    else:
        raw = oracle.synth_rows(m["corpus_seed"], 0, m["n"], m["dim"], threads=4)
        rq = oracle.synth_rows(m["query_seed"], 0, m["nq"], m["dim"])
    ids, sc = oracle.search(raw, rq, m["k"], m["dtype"], m["metric"])
    assert np.array_equal(ids, GOLD[name + "__ids"])
    assert np.array_equal(bits(sc), GOLD[name + "__score_bits"])
    # threaded scan == single-threaded scan (the cpu_baseline leg uses threads)
    ids4, sc4 = oracle.search(raw, rq, m["k"], m["dtype"], m["metric"], threads=4)
    assert np.array_equal(ids, ids4) and np.array_equal(bits(sc), bits(sc4))


def test_generator_pinned(oracle):
    assert np.array_equal(bits(oracle.synth_rows(1, 0, 3, 8)), GOLD["synth_seed1_rows0_3_dim8"])
    assert np.array_equal(bits(oracle.synth_rows(2, 123456789, 1, 5)), GOLD["synth_seed2_row123456789_dim5"])
    a = oracle.synth_rows(1, 77, 500, 96, threads=3)
    assert np.array_equal(bits(a), bits(oracle.numpy_synth_rows(1, 77, 500, 96)))
    assert np.allclose(np.linalg.norm(a.astype(np.float64), axis=1), 1.0, atol=1e-6)
    # streams with neighbouring seeds are decorrelated (seed is hashed before mixing)
    b = oracle.synth_rows(2, 77, 500, 96)
    assert abs(float((a * b).sum(1).mean())) < 0.02


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("metric", [0, 1])
def test_oracle_matches_numpy_restatement(oracle, dtype, metric):
    rng = np.random.default_rng(42 + dtype * 2 + metric)
    raw = (rng.standard_normal((3000, 70)) * rng.uniform(0.01, 100, (3000, 1))).astype(np.float32)
    rq = rng.standard_normal((6, 70)).astype(np.float32)
    pc, pq = oracle.prepare(raw, dtype, metric), oracle.prepare(rq, dtype, metric)
    assert np.array_equal(bits(pc), bits(oracle.numpy_prepare(raw, dtype, metric)))
    ids, sc = oracle.scan_topk(pc, pq, 15, metric)
    ni, ns = oracle.numpy_topk_from_scores(oracle.numpy_scores_canonical(pc, pq, metric), 15, metric)
    assert np.array_equal(ids, ni) and np.array_equal(bits(sc), bits(ns))


def test_fp64_sklearn_cross_check(oracle):
    """Independent fp64 brute force: ids agree except where the fp64 gap is below fp32
    resolution; such positions are listed and bounded, not ignored (SURVEY.md 8c rule 7)."""
    from sklearn.neighbors import NearestNeighbors
    raw = oracle.synth_rows(1, 0, 20000, 128, threads=4)
    rq = oracle.synth_rows(2, 0, 32, 128)
    k = 10
    for metric, sk_metric in ((1, "euclidean"), (0, "cosine")):
        pc, pq = oracle.prepare(raw, 0, metric), oracle.prepare(rq, 0, metric)
        ids, _ = oracle.scan_topk(pc, pq, k, metric)
        nn = NearestNeighbors(n_neighbors=k, algorithm="brute", metric=sk_metric).fit(pc.astype(np.float64))
        _, sk_ids = nn.kneighbors(pq.astype(np.float64))
        s64 = oracle.numpy_scores_fp64(pc, pq, metric)
        disagreements = []
        for qi in range(len(rq)):
            for pos in range(k):
                a, b = int(ids[qi, pos]), int(sk_ids[qi, pos])
                if a != b:
                    gap = abs(s64[qi, a] - s64[qi, b])
                    disagreements.append((qi, pos, a, b, gap))
                    assert gap < 5e-7, f"oracle and fp64 disagree beyond fp32 resolution: {(qi, pos, a, b, gap)}"
        assert len(disagreements) <= 4, disagreements
        assert [set(r) for r in ids.tolist()] == [set(r) for r in sk_ids.tolist()] or disagreements


def test_ordering_rules(oracle):
    # ties -> smaller id ; k > n -> (ID_NONE, NaN) ; zero query under cosine -> all scores 0
    x = np.array([[1, 0], [1, 0], [0, 1], [1, 0]], np.float32)
    ids, sc = oracle.search(x, np.array([[1, 0]], np.float32), 6, 0, 0)
    assert ids[0, :4].tolist() == [0, 1, 3, 2]
    assert (ids[0, 4:] == oracle.ID_NONE).all() and np.isnan(sc[0, 4:]).all()
    assert bits(sc[0, 4:]).tolist() == [0x7FC00000, 0x7FC00000]
    ids, sc = oracle.search(x, np.zeros((1, 2), np.float32), 3, 0, 0)
    assert ids[0].tolist() == [0, 1, 2] and (sc == 0).all()
    ids, sc = oracle.search(x, np.array([[1, 0]], np.float32), 2, 0, 1)   # L2: lower is better
    assert ids[0].tolist() == [0, 1] and (sc == 0).all()


def test_bf16_rounding_and_canonical_order(oracle):
    v = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.3895314e38, 1e-40], np.float32)
    r = oracle.prepare(v[None, :], 1, 1)[0]
    assert np.array_equal(bits(r), bits(oracle.numpy_bf16_round(v)))
    assert r[1] == 1.0 and r[2] == np.float32(1.015625)  # ties to even, both directions
    # canonical sum is order-sensitive: left-to-right with separate roundings
    q = np.array([1e8, 1.0, -1e8, 1.0], np.float32)
    x = np.ones(4, np.float32)
    import ctypes as C
    got = oracle.lib().orc_dot_canonical(q.ctypes.data_as(C.POINTER(C.c_float)), x.ctypes.data_as(C.POINTER(C.c_float)), 4)
    assert got == 1.0  # ((1e8 + 1) - 1e8) + 1 in fp32 = 0 + 1


def test_merge_equals_global_scan(oracle):
    rng = np.random.default_rng(5)
    raw = rng.standard_normal((3000, 48)).astype(np.float32)
    raw[1500:1510] = raw[10:20]  # duplicates across shards: merge must tie-break by global id
    rq = rng.standard_normal((7, 48)).astype(np.float32)
    for metric in (0, 1):
        full = oracle.search(raw, rq, 12, 0, metric)
        parts = [oracle.search(raw[lo:hi], rq, 12, 0, metric, id_offset=lo) for lo, hi in ((0, 1000), (1000, 2100), (2100, 3000))]
        mi, ms = oracle.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), metric)
        assert np.array_equal(mi, full[0]) and np.array_equal(bits(ms), bits(full[1]))


def test_chunked_oracle_equals_the_whole(oracle):
    """conftest.chunked_oracle_topk (the full-size GPU tests' checker: stream generated, prepared and scanned a chunk
    of rows at a time, per-chunk lists merged) = the oracle over the whole corpus, ids and score bits, both dtypes."""
    from conftest import chunked_oracle_topk
    n, dim, k = 70_001, 96, 25
    rq = oracle.synth_rows(2, 5, 4, dim)
    raw = oracle.synth_rows(1, 0, n, dim, threads=4)
    for dtype in (0, 1):
        for metric in (0, 1):
            fi, fs = oracle.search(raw, rq, k, dtype, metric, threads=4)
            ci, cs = chunked_oracle_topk(oracle, 1, n, dim, rq, k, dtype, metric, chunk=16_384, threads=4)
            assert np.array_equal(fi, ci)
            assert np.array_equal(fs.view(np.uint32), cs.view(np.uint32))
