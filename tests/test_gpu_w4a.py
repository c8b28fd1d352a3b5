"""The 4 x 1 form of the batched bf16 scan (scan_mfma_w4a_kernel: corpus rows global -> VGPR, queries in four
LDS stages, one barrier per K-tile).  It is opt-in (VROD_MFMA_W4A=1, read once per process), so the parity
cases run in a child process: ids and score bits must equal the oracle's for every K-tile count, both metrics,
dense appends, exact ties and row counts off the tile."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, ROOT_PLACEHOLDER)
import torch
import vrod_amd as va
from oracle import oracle as O
O.build()
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
def check(raw, rq, k, metric, what):
    with va.Index(raw.shape[1], "bf16", metric) as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    oi, osc = O.search(raw, rq, k, 1, {"cosine": 0, "l2": 1}[metric], threads=8)
    assert np.array_equal(ids, oi), (what, np.argwhere(ids != oi)[:4])
    assert np.array_equal(bits(sc), bits(osc)), what
    return st
n_cases = 0
for dim in (64, 100, 128, 192, 768, 1536):          # 1, 2, 2, 3, 12, 24 K-tiles per corpus tile
    for metric in ("cosine", "l2"):
        for n, nq in ((70001, 70), (300001, 300)):
            raw = O.synth_rows(11, 0, n, dim, threads=8) * np.float32(1.3)
            rq = O.synth_rows(12, 0, nq, dim)
            st = check(raw, rq, 10, metric, (dim, metric, n, nq))
            assert st["path"] == va.PATH_MFMA and st["fallback_queries"] == 0, st
            n_cases += 1
# dense appends: a small corpus where every row beats the (worst) threshold, k = 1000
raw = O.synth_rows(13, 0, 6000, 256, threads=4)
check(raw, O.synth_rows(14, 0, 130, 256), 1000, "cosine", "everything appended")
# exact ties straddling the candidate cut -> exact path for those queries, same bits
base = O.synth_rows(15, 0, 50, 128)
raw = np.concatenate([base] * 60)
st = check(raw, np.concatenate([base, base[:30]]), 25, "l2", "ties")
assert st["fallback_queries"] > 0
print("w4a ok", n_cases)
'''


@pytest.mark.gpu
def test_w4a_kernel_matches_the_oracle():
    env = dict(os.environ, VROD_MFMA_W4A="1")
    r = subprocess.run([sys.executable, "-c", CHILD.replace('ROOT_PLACEHOLDER', repr(ROOT))], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "w4a ok 24" in r.stdout
