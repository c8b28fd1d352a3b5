"""Work stealing of the 4-wave scan (kernels_mfma_w4.hip): the last 1/16 of every strip's tiles is handed out in chunks
that any work-group of the same query block may claim.  Needs strips of at least 48 tiles and K extents of at least three
K-tiles, i.e. a corpus the small parity cases never reach: 1.5M x 192 bf16, batch 1024 -- the last filtered stage walks
~73 tiles per strip.  Every (chunk, query block) pair must be scanned exactly once whoever claims it: ids and score bits
of the whole batch = the oracle's; and the same with stealing switched off (VROD_DEBUG_W4_STEAL=0, own process).
Parity unpinned by the reference (vRod holds no scan): the oracle is build-authored."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_batch_over_long_strips_equals_the_oracle(oracle, metric):
    import torch
    assert torch.cuda.is_available()
    import vrod_amd as va
    n, dim, nq, k = 1_500_000, 192, 1024, 10
    raw = oracle.synth_rows(91, 0, n, dim, threads=16)
    rq = oracle.synth_rows(92, 0, nq, dim, threads=16)
    oi, osc = oracle.search(raw, rq, k, 1, {"cosine": 0, "l2": 1}[metric], threads=16)
    with va.Index(dim, "bf16", metric) as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    assert st["path"] == va.PATH_MFMA and st["scan_launches"] >= 3, st
    assert np.array_equal(ids, oi), np.argwhere(ids != oi)[:5]
    assert np.array_equal(bits(sc), bits(osc))


def test_stealing_off_gives_the_same_bits():
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
import vrod_amd as va
ix = va.Index(192, "bf16", "cosine"); ix.add_synthetic(1, 0, 1_500_000); ix.set_path(va.PATH_MFMA)
oi = torch.empty((1024, 10), dtype=torch.int64, device="cuda"); osc = torch.empty((1024, 10), dtype=torch.float32, device="cuda")
ix.search_synthetic_device(2, 0, 1024, 10, oi, osc); torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([oi.cpu().numpy().astype(np.int64), osc.cpu().numpy().view(np.int32).astype(np.int64)], axis=1))
'''
    outs = []
    for mode in ("1", "0"):
        path = f"/tmp/vrod_steal_{os.getpid()}_{mode}.npy"
        env = dict(os.environ, VROD_DEBUG_W4_STEAL=mode)
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
        os.unlink(path)
    assert np.array_equal(outs[0], outs[1])
