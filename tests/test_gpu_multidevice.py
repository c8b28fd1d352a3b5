"""One handle over several devices (vrod_index_create with n_devices > 1, include/vrod.h).

The rows are dealt to the shards in blocks of 65536; a search runs on every shard at once and is
merged on the first device.  The 1-GPU test box exercises the whole mechanism by naming device 0
several times (each entry is a shard with its own corpus, stream and workspaces); results must be
the bits of the oracle, i.e. of a single-device index.
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


@pytest.mark.parametrize("dtype,metric,path,nq", [("f32", "cosine", 1, 3), ("bf16", "cosine", 2, 40), ("f32", "l2", 0, 1), ("bf16", "l2", 2, 300)])
def test_three_shards_equal_the_oracle(va, oracle, dtype, metric, path, nq):
    dim, n, k = 32, 3 * 65536 + 70001, 10          # four blocks + a partial one: shard sizes 131072 / 70001+65536 / 65536
    raw = oracle.synth_rows(1, 0, n, dim, threads=8) * np.float32(1.7)
    rq = oracle.synth_rows(2, 0, nq, dim)
    with va.Index(dim, dtype, metric, devices=[0, 0, 0]) as ix:
        ix.add(raw[:100000])                        # pieces that start and end inside blocks
        ix.add(raw[100000:100001])
        ix.add(raw[100001:])
        assert ix.count == n
        got = ix.get_rows(65530, 20)                # straddles a block (= shard) boundary
        ix.set_id_offset(1000)
        ix.set_path(path)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    assert np.array_equal(bits(got), bits(oracle.prepare(raw[65530:65550], DT[dtype], ME[metric])))
    oi, osc = oracle.search(raw, rq, k, DT[dtype], ME[metric], id_offset=1000)
    assert np.array_equal(ids, oi), np.argwhere(ids != oi)[:5]
    assert np.array_equal(bits(sc), bits(osc))
    assert st["nq"] == nq and st["scan_launches"] >= 3


def test_synthetic_add_small_corpus_and_device_pointers(va, oracle):
    import torch
    dim, n, k = 64, 1000, 25                        # one partial block: shards 1 and 2 stay empty, k > rows of some shards
    raw = oracle.synth_rows(7, 500, n, dim)
    rq = oracle.synth_rows(2, 0, 4, dim)
    oi, osc = oracle.search(raw, rq, k, 0, 0)
    with va.Index(dim, "f32", "cosine", devices=[0, 0, 0]) as ix:
        ix.add_synthetic(7, 500, n)
        ids, sc = ix.search(rq, k)
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))
        dq = torch.from_numpy(rq).cuda()
        di, ds = ix.search_device(dq, k)
        assert np.array_equal(di.cpu().numpy().view(np.uint64), oi) and np.array_equal(bits(ds.cpu().numpy()), bits(osc))
        with pytest.raises(va.VrodError):           # the pipelined form is per device
            ix.search_begin_device(dq, k, di, ds)
    big_k = 1200                                    # k > count: unfilled slots (ID_NONE, NaN) survive the merge
    with va.Index(dim, "f32", "l2", devices=[0, 0]) as ix:
        ix.add(raw)
        ids, sc = ix.search(rq, big_k)
    oi, osc = oracle.search(raw, rq, big_k, 0, 1)
    assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))


def test_rejected_add_rolls_every_shard_back(va, oracle):
    dim = 16
    raw = oracle.synth_rows(1, 0, 140000, dim, threads=8)
    bad = raw[70000:140000].copy()
    bad[69000, 3] = np.inf                          # lands in the third block -> third shard's piece
    rq = oracle.synth_rows(2, 0, 2, dim)
    with va.Index(dim, "f32", "cosine", devices=[0, 0, 0]) as ix:
        ix.add(raw[:70000])
        with pytest.raises(va.VrodError) as e:
            ix.add(bad)
        assert e.value.code == 2 and ix.count == 70000
        ix.add(raw[70000:])                         # the same rows, clean: the handle is consistent
        ids, sc = ix.search(rq, 10)
    oi, osc = oracle.search(raw, rq, 10, 0, 0)
    assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))
