"""N>1 path on CPU: world_size-2 gloo run of the sharding + all-gather + merge plumbing
(vrod_amd/shard.py, the same functions bench.py drives over RCCL).  The local scans here are
the oracle's (no GPU in this container); what is under test is the exchange and the merge."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, dim, nq, k, metric, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from vrod_amd.shard import all_gather_packed, all_gather_topk, alloc_packed, shard_range, unpack_gathered
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    raw = O.synth_rows(1, lo, hi - lo, dim)           # this rank's rows of the shared stream
    rq = O.synth_rows(2, 0, nq, dim)
    ids, sc = O.search(raw, rq, k, 0, metric, id_offset=lo)
    gi, gs = all_gather_topk(dist, torch.from_numpy(ids.view(np.int64)), torch.from_numpy(sc))
    mi, ms = O.merge_topk(gi.numpy().view(np.uint64), gs.numpy(), metric)
    # the packed single-collective form bench.py uses (ids | scores in one byte block per rank)
    packed, pi, ps = alloc_packed(nq, k, "cpu")
    pi.copy_(torch.from_numpy(ids.view(np.int64)))
    ps.copy_(torch.from_numpy(sc))
    ui, us = unpack_gathered(all_gather_packed(dist, packed), world, nq, k)
    assert np.array_equal(ui, gi.numpy().view(np.uint64)) and np.array_equal(us.view(np.uint32), gs.numpy().view(np.uint32))
    if rank == 0:
        q.put((mi, ms))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric", [0, 1])
def test_world2_shard_gather_merge_equals_single(oracle, metric):
    import torch.multiprocessing as mp
    n_total, dim, nq, k, world = 5001, 64, 6, 10, 2   # odd row count: uneven shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, dim, nq, k, metric, q)) for r in range(world)]
    for p in procs:
        p.start()
    mi, ms = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    raw = oracle.synth_rows(1, 0, n_total, dim)
    rq = oracle.synth_rows(2, 0, nq, dim)
    oi, osc = oracle.search(raw, rq, k, 0, metric)
    assert np.array_equal(mi, oi)
    assert np.array_equal(ms.view(np.uint32), osc.view(np.uint32))


def test_shard_ranges_cover_everything():
    from vrod_amd.shard import shard_range
    for n in (0, 1, 7, 10_000_000, 40_000_000):
        for w in (1, 2, 3, 4, 8):
            r = [shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
