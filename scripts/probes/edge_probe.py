"""One-off sweep of awkward shapes against the oracle (dev tool)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import vrod_amd as va
from oracle import oracle as O
O.build()
rng = np.random.default_rng(9)
cases = []
for n in (8191, 8192, 8193, 8447, 8448, 16384, 16385, 65535, 65537):
    cases.append((n, 64, 300, 10, "bf16", "cosine", 2, 0))
    cases.append((n, 100, 40, 100, "f32", "l2", 2, 0))
cases += [(50000, 96, 300, 3584, "bf16", "cosine", 2, 0), (50000, 96, 40, 3584, "f32", "l2", 2, 0), (50000, 96, 9, 3584, "f32", "cosine", 1, 0),
          (30000, 1, 300, 10, "bf16", "cosine", 2, 0), (30000, 1, 300, 10, "bf16", "l2", 2, 0), (30000, 2, 40, 10, "f32", "cosine", 2, 0),
          (20000, 64, 1023, 10, "bf16", "cosine", 2, 1 << 40), (20000, 64, 1025, 10, "bf16", "l2", 2, (1 << 62) + 5)]
bad = 0
for (n, dim, nq, k, dtype, metric, path, off) in cases:
    raw = rng.standard_normal((n, dim)).astype(np.float32)
    rq = rng.standard_normal((nq, dim)).astype(np.float32)
    try:
        with va.Index(dim, dtype, metric) as ix:
            ix.add(raw); ix.set_id_offset(off); ix.set_path(path)
            ids, sc = ix.search(rq, k); st = ix.last_stats()
        oi, osc = O.search(raw, rq, k, 0 if dtype == "f32" else 1, 0 if metric == "cosine" else 1, id_offset=off, threads=8)
        ok = np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
        res = "ok" if ok else "MISMATCH"
    except Exception as e:
        res = "ERROR " + str(e)[:100]; st = {}
    bad += res != "ok"
    print(f"n={n} d={dim} nq={nq} k={k} {dtype} {metric} p{path} off={off}: {res} fb={st.get('fallback_queries')}", flush=True)
# degenerate contents
for name, raw in (("all-zero rows", np.zeros((20000, 64), np.float32)), ("all-equal rows", np.ones((20000, 64), np.float32)),
                  ("two distinct rows", np.repeat(rng.standard_normal((2, 64)).astype(np.float32), 10000, axis=0))):
    rq = rng.standard_normal((40, 64)).astype(np.float32)
    for dtype, metric, path in (("bf16", "cosine", 2), ("f32", "l2", 2), ("f32", "cosine", 1)):
        try:
            with va.Index(64, dtype, metric) as ix:
                ix.add(raw); ix.set_path(path)
                ids, sc = ix.search(rq[: (4 if path == 1 else 40)], 10); st = ix.last_stats()
            oi, osc = O.search(raw, rq[: (4 if path == 1 else 40)], 10, 0 if dtype == "f32" else 1, 0 if metric == "cosine" else 1, threads=8)
            ok = np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
            res = "ok" if ok else "MISMATCH"
        except Exception as e:
            res = "ERROR " + str(e)[:100]; st = {}
        bad += res != "ok"
        print(f"{name} {dtype} {metric} p{path}: {res} fb={st.get('fallback_queries')}", flush=True)
print("BAD:", bad)
