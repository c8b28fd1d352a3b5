"""Generates tests/golden/golden.npz from the CPU oracle (oracle/vrod_oracle.c).

vRod ships no tests or fixtures for this path ("parity unpinned"), so the
golden vectors are build-authored: synthetic-stream seeds + expected ids + expected score
BITS.  Every case is cross-checked here against the independent numpy restatement
(oracle.numpy_*: fp32 canonical order, different code path) before it is written.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

# name, n, dim, dtype, metric, nq, k   
CASES = [
    ("cfg1_10k_128_f32_cos_q1_k10", 10000, 128, 0, 0, 1, 10),
    ("4096_768_f32_l2_q4_k100", 4096, 768, 0, 1, 4, 100),
    ("8192_768_bf16_cos_q64_k10", 8192, 768, 1, 0, 64, 10),
    ("4096_1536_f32_cos_q8_k1000", 4096, 1536, 0, 0, 8, 1000),
    ("odd_1000_100_bf16_l2_q3_k7", 1000, 100, 1, 1, 3, 7),
    ("k_equals_n_37_33_f32_l2", 37, 33, 0, 1, 2, 37),
    ("k_gt_n_20_40_f32_cos", 20, 40, 0, 0, 2, 25),
    ("k1_500_64_bf16_cos", 500, 64, 1, 0, 5, 1),
]


def adversarial():
    """Duplicated rows (tie-break by id) + a zero vector, raw data stored in the fixture."""
    rng = np.random.default_rng(1234)
    base = rng.standard_normal((16, 24)).astype(np.float32)
    raw = np.concatenate([base, base, base[::-1]])  # 48 rows, every vector 3 times
    raw[5] = 0.0
    rq = np.concatenate([base[:2] * 2.0, np.zeros((1, 24), np.float32)])
    return raw, rq


def main():
    out = {}
    meta = {}
    for name, n, dim, dt, me, nq, k in CASES:
        raw = O.synth_rows(1, 0, n, dim, threads=4)
        rq = O.synth_rows(2, 0, nq, dim)
        ids, sc = O.search(raw, rq, k, dt, me)
        # independent restatement
        pc, pq = O.numpy_prepare(raw, dt, me), O.numpy_prepare(rq, dt, me)
        ni, ns = O.numpy_topk_from_scores(O.numpy_scores_canonical(pc, pq, me), k, me)
        assert np.array_equal(ids, ni) and np.array_equal(sc.view(np.uint32), ns.view(np.uint32)), name
        out[name + "__ids"] = ids
        out[name + "__score_bits"] = sc.view(np.uint32)
        meta[name] = dict(n=n, dim=dim, dtype=dt, metric=me, nq=nq, k=k, corpus_seed=1, query_seed=2)
    raw, rq = adversarial()
    for me, mname in ((0, "cos"), (1, "l2")):
        name = f"adversarial_dups_zero_{mname}"
        ids, sc = O.search(raw, rq, 9, 0, me)
        pc, pq = O.numpy_prepare(raw, 0, me), O.numpy_prepare(rq, 0, me)
        ni, ns = O.numpy_topk_from_scores(O.numpy_scores_canonical(pc, pq, me), 9, me)
        assert np.array_equal(ids, ni) and np.array_equal(sc.view(np.uint32), ns.view(np.uint32)), name
        out[name + "__ids"] = ids
        out[name + "__score_bits"] = sc.view(np.uint32)
        out[name + "__raw"] = raw
        out[name + "__queries"] = rq
        meta[name] = dict(dtype=0, metric=me, k=9, raw=True)
    # a few raw synthetic values pin the generator itself
    out["synth_seed1_rows0_3_dim8"] = O.synth_rows(1, 0, 3, 8).view(np.uint32)
    out["synth_seed2_row123456789_dim5"] = O.synth_rows(2, 123456789, 1, 5).view(np.uint32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden.npz"), **out)
    with open(os.path.join(ROOT, "tests", "golden", "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
