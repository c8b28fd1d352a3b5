import sys
sys.path.insert(0, ".")
import numpy as np
import vrod_amd as va
from oracle import oracle as O
O.build()
rng = np.random.default_rng(5)
for dim in [int(x) for x in (sys.argv[1:] or ["4096", "12288", "20000", "40000", "65536"])]:
    n = 3000 if dim <= 20000 else 600
    raw = rng.standard_normal((n, dim)).astype(np.float32)
    for dtype in ("f32", "bf16"):
        for metric in ("cosine", "l2"):
            for path, nq in ((1, 2), (2, 9), (3, 1)):
                rq = rng.standard_normal((nq, dim)).astype(np.float32)
                try:
                    with va.Index(dim, dtype, metric) as ix:
                        ix.add(raw); ix.set_path(path)
                        ids, sc = ix.search(rq, 5)
                    oi, osc = O.search(raw, rq, 5, 0 if dtype == "f32" else 1, 0 if metric == "cosine" else 1, threads=8)
                    ok = np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
                    print(f"dim {dim} {dtype} {metric} path {path}: {'ok' if ok else 'MISMATCH'}", flush=True)
                except Exception as e:
                    print(f"dim {dim} {dtype} {metric} path {path}: ERROR {str(e)[:120]}", flush=True)
