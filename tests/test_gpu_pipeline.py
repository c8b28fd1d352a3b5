"""GPU tests of the pipelined search (vrod_search_begin_* / vrod_search_end, include/vrod.h).

Two searches may be in flight on one handle.  The results of a pipelined run must be the
bits of the strictly sequential run and of the CPU oracle, including when the search that is
completed LATE needs the exact path (its prepared queries must have survived the next
search's prepare launch) and when a batch with NaN queries sits next to a good one.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = {"f32": 0, "bf16": 1}
ME = {"cosine": 0, "l2": 1}


@pytest.fixture(scope="module")
def va():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    import vrod_amd
    vrod_amd.load()
    return vrod_amd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def pipelined(ix, torch, batches, k):
    """begin(0); for s: begin(s+1); end(s) -- the bench.py loop.  Returns [(ids, scores, stats)]."""
    dev = torch.device("cuda", 0)
    dq = [torch.from_numpy(b).to(dev) for b in batches]
    outs = [(torch.empty((b.shape[0], k), dtype=torch.int64, device=dev),
             torch.empty((b.shape[0], k), dtype=torch.float32, device=dev)) for b in batches]
    res = []
    ix.search_begin_device(dq[0], k, *outs[0])
    for s in range(len(batches)):
        if s + 1 < len(batches):
            ix.search_begin_device(dq[s + 1], k, *outs[s + 1])
            assert ix.pending == 2
        ix.search_end()
        st = ix.last_stats()
        res.append((outs[s][0].cpu().numpy().view(np.uint64), outs[s][1].cpu().numpy(), st))
    assert ix.pending == 0
    return res


@pytest.mark.parametrize("dtype,metric,path,nq,split", [("bf16", "cosine", 2, 300, None), ("f32", "l2", 1, 3, None),
                                                        ("f32", "cosine", 2, 40, "0"), ("f32", "cosine", 2, 40, None),
                                                        ("bf16", "l2", 1, 8, None), ("f32", "cosine", 3, 2, None)])
def test_pipelined_equals_sequential_and_oracle(va, oracle, dtype, metric, path, nq, split):
    import torch
    from conftest import f32_split
    raw = oracle.synth_rows(1, 0, 30000, 256, threads=8)
    # batches of different sizes: slot buffers regrow while the other slot is in flight
    sizes = [nq, max(1, nq // 2), nq + 5, nq]
    batches = [oracle.synth_rows(2, 1000 * i, n, 256) for i, n in enumerate(sizes)]
    k = 10
    with f32_split(split), va.Index(256, dtype, metric) as ix:
        ix.add(raw)
        ix.set_path(path)
        seq = [ix.search(b, k) for b in batches]
        pip = pipelined(ix, torch, batches, k)
    with f32_split(split), va.Index(256, dtype, metric) as ix:
        ix.add(raw)
        ix.set_path(path)
        pip_first = pipelined(ix, torch, batches, k)   # an fp32 handle builds its bf16 planes inside the first pipelined batch
    for a, b in zip(pip, pip_first):
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    for i, b in enumerate(batches):
        oi, osc = oracle.search(raw, b, k, DT[dtype], ME[metric])
        ids, sc, st = pip[i]
        assert st["nq"] == b.shape[0] and st["path"] == path
        assert np.array_equal(ids, seq[i][0]) and np.array_equal(bits(sc), bits(seq[i][1])), f"batch {i}: pipelined != sequential"
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc)), f"batch {i}: pipelined != oracle"


@pytest.mark.parametrize("path", [1, 2])
def test_late_completion_takes_exact_path_with_its_own_queries(va, oracle, path):
    """Massive exact ties defeat the certificate: the exact path of batch s runs after batch
    s+1 (different queries) has been enqueued, and must still use batch s's queries."""
    import torch
    rng = np.random.default_rng(5)
    base = rng.standard_normal((20, 64)).astype(np.float32)
    raw = np.concatenate([base] * 100)   # 100 copies > k': the cut always falls inside a tie group
    batches = [base[3 * i:3 * i + 3] + 0.01 * rng.standard_normal((3, 64)).astype(np.float32) for i in range(4)]
    k = 25
    with va.Index(64, "f32", "cosine") as ix:
        ix.add(raw)
        ix.set_path(path)
        pip = pipelined(ix, torch, batches, k)
    fallbacks = 0
    for i, b in enumerate(batches):
        oi, osc = oracle.search(raw, b, k, 0, 0)
        ids, sc, st = pip[i]
        fallbacks += st["fallback_queries"]
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc)), f"batch {i}"
    assert fallbacks > 0, "the case is meant to exercise the exact path"


def test_nan_batch_does_not_poison_its_neighbour(va, oracle):
    import torch
    dev = torch.device("cuda", 0)
    raw = oracle.synth_rows(1, 0, 5000, 96)
    good = oracle.synth_rows(2, 0, 4, 96)
    bad = good.copy()
    bad[1, 7] = np.nan
    k = 10
    oi, osc = oracle.search(raw, good, k, 0, 0)
    with va.Index(96, "f32", "cosine") as ix:
        ix.add(raw)
        dgood, dbad = torch.from_numpy(good).to(dev), torch.from_numpy(bad).to(dev)
        o = [(torch.empty((4, k), dtype=torch.int64, device=dev), torch.empty((4, k), dtype=torch.float32, device=dev)) for _ in range(3)]
        ix.search_begin_device(dbad, k, *o[0])
        ix.search_begin_device(dgood, k, *o[1])
        with pytest.raises(va.VrodError) as e:
            ix.search_end()
        assert e.value.code == 2
        ix.search_begin_device(dbad, k, *o[2])
        ix.search_end()          # the good batch between two bad ones
        assert np.array_equal(o[1][0].cpu().numpy().view(np.uint64), oi)
        assert np.array_equal(bits(o[1][1].cpu().numpy()), bits(osc))
        with pytest.raises(va.VrodError):
            ix.search_end()
        assert ix.pending == 0
        ids, sc = ix.search(good, k)   # and the handle is healthy afterwards
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))


def test_pipeline_protocol_errors(va, oracle):
    import torch
    dev = torch.device("cuda", 0)
    raw = oracle.synth_rows(1, 0, 2000, 64)
    q = torch.from_numpy(oracle.synth_rows(2, 0, 2, 64)).to(dev)
    k = 5
    o = [(torch.empty((2, k), dtype=torch.int64, device=dev), torch.empty((2, k), dtype=torch.float32, device=dev)) for _ in range(3)]
    with va.Index(64, "f32", "l2") as ix:
        ix.add(raw)
        with pytest.raises(va.VrodError):       # nothing pending
            ix.search_end()
        ix.search_begin_device(q, k, *o[0])
        ix.search_begin_device(q, k, *o[1])
        with pytest.raises(va.VrodError):       # a third one does not fit
            ix.search_begin_device(q, k, *o[2])
        with pytest.raises(va.VrodError):       # the corpus is frozen while searches are pending
            ix.add(raw[:10])
        with pytest.raises(va.VrodError):       # so is the synchronous search
            ix.search(raw[:1], k)
        assert ix.pending == 2
        ix.search_end()
        ix.search_end()
        assert ix.count == 2000
        assert torch.equal(o[0][0], o[1][0]) and torch.equal(o[0][1], o[1][1])
        # synthetic-query form: same stream as the oracle's generator
        ix.search_begin_synthetic_device(2, 0, 2, k, *o[2])
        ix.search_end()
        assert torch.equal(o[2][0], o[0][0])
    # destroying a handle with a search still pending must not hang or crash
    ix = va.Index(64, "f32", "l2")
    ix.add(raw)
    ix.search_begin_device(q, k, *o[0])
    ix.close()


def test_more_query_blocks_than_work_group_slots(va, oracle):
    """nq > 256 * 32: the 4-wave MFMA kernel takes one query block per work-group, so the launcher
    splits the batch into several launches (33 query blocks on a 256-CU device)."""
    raw = oracle.synth_rows(1, 0, 20000, 64, threads=8)
    rq = oracle.synth_rows(2, 0, 8300, 64, threads=8)
    k = 5
    with va.Index(64, "bf16", "cosine") as ix:
        ix.add(raw)
        ix.set_path(2)
        ids, sc = ix.search(rq, k)
        assert ix.last_stats()["path"] == 2
    oi, osc = oracle.search(raw, rq, k, 1, 0)
    assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))


def test_overlap_of_two_scan_launches_is_reported_once(va):
    """Pipelined staged MFMA searches over a shard-sized corpus put the next batch's sample launch beside this batch's
    last stage; vrod_search_stats.overlap_ms is how long both were in flight, so that scan_ms - overlap_ms is the time
    at least one scan launch ran (what bench.py's roofline divides by).  Synchronous searches overlap nothing."""
    import torch
    dev = torch.device("cuda", 0)
    nq, k = 512, 10
    with va.Index(128, "bf16", "cosine") as ix:
        ix.add_synthetic(1, 0, 600000)
        ix.set_profiling(True)
        outs = [(torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev)) for _ in range(2)]
        stats = []
        ix.search_begin_synthetic_device(2, 0, nq, k, *outs[0])
        for s in range(6):
            if s + 1 < 6:
                ix.search_begin_synthetic_device(2, (s + 1) * nq, nq, k, *outs[(s + 1) % 2])
            ix.search_end()
            stats.append(ix.last_stats())
        ix.search_synthetic_device(2, 0, nq, k, *outs[0])
        sync = ix.last_stats()
    assert all(st["path"] == 2 and st["scan_launches"] >= 3 and st["sample_ms"] > 0 for st in stats), stats
    assert stats[0]["overlap_ms"] == 0                      # nothing in front of the first search
    for st in stats[1:]:
        assert 0 <= st["overlap_ms"] <= st["sample_ms"] + 1e-3 and st["overlap_ms"] < st["scan_ms"], st
    # (whether the two launches really meet is a race between a ~10-us sample launch and the ~10-us compaction in front of the
    #  last stage: at this size they usually do, at shard sizes always -- bench.py's roofline.overlap shows 0.6 ms per batch)
    assert sync["overlap_ms"] == 0 and sync["sample_ms"] > 0
