"""One handle over several devices (vrod_index_create with n_devices > 1, include/vrod.h).

The rows are dealt to the shards in blocks of 65536; a search runs on every shard at once, the
per-shard lists are exchanged with one RCCL all-gather per distinct device and merged on the first
device.  The 1-GPU test box exercises the whole mechanism by naming device 0 several times (each
entry is a shard with its own corpus, stream and workspaces; the communicator then has ONE rank and
that device contributes a block of three lists); results must be the bits of the oracle, i.e. of a
single-device index.  VROD_RCCL=0 runs the same searches over peer copies.
"""
import contextlib
import os
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


@contextlib.contextmanager
def rccl(mode):
    """VROD_RCCL for the handles created inside (read at vrod_index_create): None = default (RCCL)."""
    old = os.environ.get("VROD_RCCL")
    if mode is None:
        os.environ.pop("VROD_RCCL", None)
    else:
        os.environ["VROD_RCCL"] = mode
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("VROD_RCCL", None)
        else:
            os.environ["VROD_RCCL"] = old


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
    assert st["exchange"] == 1                      # the lists travelled through ncclAllGather (1-rank communicator here)


@pytest.mark.parametrize("mode,kind", [(None, 1), ("0", 2)])
@pytest.mark.parametrize("nq,k", [(1, 5), (3, 25), (2, 7)])
def test_odd_nq_times_k_blocks_stay_aligned(va, oracle, mode, kind, nq, k):
    """A shard's packed block is nq*k*12 bytes: with nq*k odd the next list used to start 4 bytes off an
    8-byte stride (ids of every shard but the first were misread).  Blocks are padded to 16 bytes now."""
    dim, n = 48, 2 * 65536 + 12345
    raw = oracle.synth_rows(3, 0, n, dim, threads=8)
    rq = oracle.synth_rows(2, 0, nq, dim)
    with rccl(mode), va.Index(dim, "f32", "l2", devices=[0, 0, 0]) as ix:
        ix.add(raw)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    oi, osc = oracle.search(raw, rq, k, 0, 1)
    assert np.array_equal(ids, oi), (ids, oi)
    assert np.array_equal(bits(sc), bits(osc))
    assert st["exchange"] == kind


def test_pipelined_searches_on_a_multi_device_handle(va, oracle):
    """begin(s+1) before end(s) on ONE handle over several shards: the exchange + merge of batch s runs
    while every shard scans batch s+1; results land in the caller's device buffers in FIFO order."""
    import torch
    dim, n, nq, k = 64, 3 * 65536 + 999, 40, 10
    raw = oracle.synth_rows(1, 0, n, dim, threads=8)
    with va.Index(dim, "bf16", "cosine", devices=[0, 0]) as ix:
        ix.add_synthetic(1, 0, n)
        ix.set_path(va.PATH_MFMA)
        outs = [(torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda")) for _ in range(2)]
        got = []
        steps = 5
        ix.search_begin_synthetic_device(2, 0, nq, k, *outs[0])
        for s in range(steps):
            if s + 1 < steps:
                if s % 2 == 0:      # alternate the two begin forms
                    dq = va.synth_rows_device(0, 2, (s + 1) * nq, nq, dim)
                    ix.search_begin_device(dq, k, *outs[(s + 1) % 2])
                else:
                    ix.search_begin_synthetic_device(2, (s + 1) * nq, nq, k, *outs[(s + 1) % 2])
                assert ix.pending == 2
                with pytest.raises(va.VrodError):       # a third one must wait
                    ix.search_begin_synthetic_device(2, 0, nq, k, *outs[0])
            ix.search_end()
            got.append((outs[s % 2][0].cpu().numpy().view(np.uint64).copy(), outs[s % 2][1].cpu().numpy().copy()))
        assert ix.pending == 0
        with pytest.raises(va.VrodError):
            ix.search_end()
    for s in range(steps):
        rq = oracle.synth_rows(2, s * nq, nq, dim)
        oi, osc = oracle.search(raw, rq, k, 1, 0, threads=8)
        assert np.array_equal(got[s][0], oi), s
        assert np.array_equal(bits(got[s][1]), bits(osc)), s


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
        ix.search_begin_device(dq, k, di, ds)       # the pipelined form, one batch
        ix.search_end()
        assert np.array_equal(di.cpu().numpy().view(np.uint64), oi) and np.array_equal(bits(ds.cpu().numpy()), bits(osc))
        per = [ix.shard_stats(g) for g in range(3)]  # per-shard counters of that search
        assert all(p["device"] == 0 and p["nq"] in (0, 4) for p in per) and per[0]["nq"] == 4 and per[0]["scan_launches"] >= 1
        assert sum(p["scan_launches"] for p in per) == ix.last_stats()["scan_launches"]
        with pytest.raises(va.VrodError):
            ix.shard_stats(3)
    big_k = 1200                                    # k > count: unfilled slots (ID_NONE, NaN) survive the merge
    with va.Index(dim, "f32", "l2", devices=[0, 0]) as ix:
        ix.add(raw)
        ids, sc = ix.search(rq, big_k)
    oi, osc = oracle.search(raw, rq, big_k, 0, 1)
    assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))


def test_shard_stats_of_a_single_device_handle_are_its_own(va, oracle):
    raw = oracle.synth_rows(3, 0, 5000, 64)
    with va.Index(64, "f32", "l2") as ix:
        ix.add(raw)
        ix.search(oracle.synth_rows(2, 0, 3, 64), 5)
        st, sh = ix.last_stats(), ix.shard_stats(0)
        assert sh.pop("device") == 0 and sh == st
        with pytest.raises(va.VrodError):
            ix.shard_stats(1)


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


def test_merge_waits_for_the_callers_earlier_use_of_the_output_buffers(va, oracle):
    """The merge of a multi-device handle writes the caller's output buffers on the library's exchange stream: it must be
    ordered behind what the caller's stream still does with them.  A slow consumer of the previous batch's results is put
    on a torch stream, the next search is begun on that stream into the SAME buffers: the consumer must have seen the
    previous batch's ids, not the next batch's (the single-device path has the same ordering)."""
    import torch
    dim, n, nq, k = 64, 150_000, 64, 10
    raw = oracle.synth_rows(61, 0, n, dim, threads=8)
    dev = torch.device("cuda", 0)
    with va.Index(dim, "bf16", "cosine", devices=[0, 0]) as ix:
        ix.add(raw)
        oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        osc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            ix.search_synthetic_device(62, 0, nq, k, oi, osc)          # batch A, synchronous: results in oi
            want = oi.clone()
            # a long-running kernel chain on the caller's stream, then the consumer of batch A's ids
            junk = torch.empty(64 << 20, dtype=torch.float32, device=dev)
            for _ in range(40):
                junk.normal_()
            seen = oi.clone()
            ix.search_begin_synthetic_device(62, nq, nq, k, oi, osc)   # batch B into the same buffers, begun behind the consumer
            ix.search_end()
        st.synchronize()
        torch.cuda.synchronize(dev)
        assert torch.equal(seen, want), "the merge of batch B overwrote the buffers before the caller's stream had read batch A"
        assert not torch.equal(oi, want)


def test_a_device_that_does_not_exist_is_an_error_not_an_abort(va):
    with pytest.raises(Exception) as e:
        va.Index(32, "bf16", "cosine", devices=[0, 99])
    assert "device 99 out of range" in str(e.value)
    with pytest.raises(Exception) as e:
        va.Index(32, "bf16", "cosine", device=99)
    assert "out of range" in str(e.value)


def test_without_librccl_the_exchange_falls_back_to_peer_copies(va, oracle):
    """VROD_RCCL_LIB pointing at something that is no library: a one-time warning, stats.exchange = 2, same bits.
    (Own process: the library binds librccl once per process.)"""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, ".")
import vrod_amd as va
from oracle import oracle as O
raw = O.synth_rows(71, 0, 140000, 48, threads=4); rq = O.synth_rows(72, 0, 5, 48)
oi, osc = O.search(raw, rq, 7, 1, 0, threads=4)
with va.Index(48, "bf16", "cosine", devices=[0, 0, 0]) as ix:
    ix.add(raw)
    ids, sc = ix.search(rq, 7)
    st = ix.last_stats()
assert st["exchange"] == 2, st
assert np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VROD_RCCL_LIB="/etc/hostname")
    env.pop("VROD_RCCL", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
    assert "librccl could not be loaded" in r.stderr


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 1]])
def test_distinct_devices_through_rccl_and_peer_copies(va, oracle, devices):
    """Two physical devices (skipped on the 1-GPU box): the in-process RCCL all-gather over a communicator of two ranks, an
    uneven group ([0, 1, 1]: the first device contributes a filler list), against VROD_RCCL=0 and the oracle."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    dim, n, nq, k = 96, 400_000, 33, 9
    raw = oracle.synth_rows(81, 0, n, dim, threads=8)
    rq = oracle.synth_rows(82, 0, nq, dim)
    oi, osc = oracle.search(raw, rq, k, 1, 0, threads=8)
    for mode, kind in ((None, 1), ("0", 2)):
        with rccl(mode), va.Index(dim, "bf16", "cosine", devices=devices) as ix:
            ix.add(raw)
            ids, sc = ix.search(rq, k)
            st = ix.last_stats()
        assert st["exchange"] == kind, st
        assert np.array_equal(ids, oi) and np.array_equal(bits(sc), bits(osc))
