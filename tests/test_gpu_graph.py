"""hipGraph replay of small, launch-bound searches (vrod_index.hip search_enqueue): when a slot
sees the same search again -- same pointers, sizes, corpus, workspaces -- while another search is
pending (a host-bound pipeline), its launches are captured and then replayed as one graph launch.
The replays must read the CURRENT queries and corpus contents, report errors like the plain path,
and step aside as soon as anything differs.
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


@pytest.mark.parametrize("dtype,metric,nq", [("f32", "cosine", 1), ("f32", "l2", 4), ("bf16", "cosine", 3)])
def test_repeated_small_searches_replay_correctly(va, oracle, dtype, metric, nq):
    DT = {"f32": 0, "bf16": 1}
    ME = {"cosine": 0, "l2": 1}
    n, dim, k = 10000, 128, 10
    raw = oracle.synth_rows(1, 0, n, dim)
    with va.Index(dim, dtype, metric) as ix:
        ix.add(raw)
        for rep in range(10):                       # slots alternate: each sees plain, capture, then replays
            rq = oracle.synth_rows(2, 100 * rep, nq, dim)
            ids, sc = ix.search(rq, k)
            oi, osc = oracle.search(raw, rq, k, DT[dtype], ME[metric])
            assert np.array_equal(ids, oi), f"repetition {rep}"
            assert np.array_equal(bits(sc), bits(osc)), f"repetition {rep}"
        # a bad query through the replayed graph is reported, and the next search is clean
        bad = oracle.synth_rows(2, 0, nq, dim)
        bad[0, 5] = np.nan
        for _ in range(2):
            with pytest.raises(va.VrodError) as e:
                ix.search(bad, k)
            assert e.value.code == 2
        rq = oracle.synth_rows(2, 7, nq, dim)
        ids, sc = ix.search(rq, k)
        oi, osc = oracle.search(raw, rq, k, DT[dtype], ME[metric])
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))
        # the corpus grows: the captured graph no longer applies (and is re-captured later)
        more = oracle.synth_rows(3, 0, 500, dim)
        ix.add(more)
        full = np.concatenate([raw, more])
        for rep in range(6):
            rq = oracle.synth_rows(2, 900 + rep, nq, dim)
            ids, sc = ix.search(rq, k)
            oi, osc = oracle.search(full, rq, k, DT[dtype], ME[metric])
            assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc)), f"after add, repetition {rep}"
        # a different k between identical searches
        ids, sc = ix.search(rq, 3)
        oi, osc = oracle.search(full, rq, 3, DT[dtype], ME[metric])
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))


def test_pipelined_small_searches_replay_correctly(va, oracle):
    """The case the replay is for: two small searches in flight, the same buffers over and over."""
    import torch
    dev = torch.device("cuda", 0)
    n, dim, k, nq = 10000, 128, 10, 2
    raw = oracle.synth_rows(1, 0, n, dim)
    with va.Index(dim, "f32", "cosine") as ix:
        ix.add(raw)
        q = [torch.empty((nq, dim), dtype=torch.float32, device=dev) for _ in range(2)]
        o = [(torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev)) for _ in range(2)]
        steps = 12
        host_q = [oracle.synth_rows(2, 50 * s, nq, dim) for s in range(steps)]
        results = []
        q[0].copy_(torch.from_numpy(host_q[0])); torch.cuda.synchronize()
        ix.search_begin_device(q[0], k, *o[0])
        for s in range(steps):
            if s + 1 < steps:
                q[(s + 1) % 2].copy_(torch.from_numpy(host_q[s + 1])); torch.cuda.synchronize()
                ix.search_begin_device(q[(s + 1) % 2], k, *o[(s + 1) % 2])
            ix.search_end()
            results.append((o[s % 2][0].cpu().numpy().view(np.uint64).copy(), o[s % 2][1].cpu().numpy().copy()))
        for s in range(steps):
            oi, osc = oracle.search(raw, host_q[s], k, 0, 0)
            assert np.array_equal(results[s][0], oi), f"step {s}"
            assert np.array_equal(bits(results[s][1]), bits(osc)), f"step {s}"
