"""The sample pass of the batched scan has two forms on the 2 x 2 4-wave kernel: every score of the sample rows
written out (threshold = exact j-th best), or one score per group of 32 rows (threshold = j-th best of the group
bests: a valid lower bound, 1/32 of the writes; default where the groups outnumber j by 8x).  Both must give the
oracle's bits; VROD_DEBUG_SAMPLE_GROUPED is read once per process, so each form runs in a child process.
Parity unpinned by the reference (vRod holds no scan): the oracle is build-authored."""
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
n_cases = 0
for dtype, metric, dim, k in (("bf16", "cosine", 768, 10), ("bf16", "l2", 192, 10), ("f32", "cosine", 128, 10), ("bf16", "cosine", 64, 40)):
    raw = O.synth_rows(51, 0, 600_000, dim, threads=8) * np.float32(1.7)
    rq = O.synth_rows(52, 0, 300, dim)
    # planted near-duplicates inside the sample rows: the j best of the sample then share groups of 32 rows
    raw[1000:1016] = rq[:16] + np.float32(1e-3) * raw[1000:1016]
    with va.Index(dim, dtype, metric) as ix:
        ix.add(raw)
        ix.set_path(va.PATH_MFMA)
        ids, sc = ix.search(rq, k)
        st = ix.last_stats()
    oi, osc = O.search(raw, rq, k, {"f32": 0, "bf16": 1}[dtype], {"cosine": 0, "l2": 1}[metric], threads=8)
    assert st["path"] == va.PATH_MFMA and st["scan_launches"] >= 3, st
    assert np.array_equal(ids, oi), (dtype, metric, np.argwhere(ids != oi)[:4])
    assert np.array_equal(bits(sc), bits(osc)), (dtype, metric)
    n_cases += 1
print("sample form ok", n_cases)
'''


@pytest.mark.gpu
@pytest.mark.parametrize("grouped", ["1", "0"])
def test_both_sample_forms_match_the_oracle(grouped):
    env = dict(os.environ, VROD_DEBUG_SAMPLE_GROUPED=grouped)
    r = subprocess.run([sys.executable, "-c", CHILD.replace('ROOT_PLACEHOLDER', repr(ROOT))], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "sample form ok 4" in r.stdout
