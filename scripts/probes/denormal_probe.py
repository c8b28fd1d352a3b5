import sys, os
sys.path.insert(0, ".")
import numpy as np
import vrod_amd as va
from oracle import oracle as O
O.build()
rng = np.random.default_rng(3)
for scale in [float(x) for x in (sys.argv[1:] or ["1e-18", "1e-19", "3e-20", "1e-21"])]:
    for dtype, split in [("f32", False), ("f32", True), ("bf16", False)]:
        for metric in ["l2", "cosine"]:
            os.environ["VROD_F32_SPLIT"] = "1" if split else "0"
            n, dim, nq, k = 20000, 96, 40, 10
            raw = (rng.standard_normal((n, dim)) * scale).astype(np.float32)
            rq = (rng.standard_normal((nq, dim)) * scale).astype(np.float32)
            with va.Index(dim, dtype, metric) as ix:
                ix.add(raw); ix.set_path(2)
                ids, sc = ix.search(rq, k); st = ix.last_stats()
            oi, osc = O.search(raw, rq, k, 0 if dtype == "f32" else 1, 0 if metric == "cosine" else 1)
            ok = np.array_equal(ids, oi) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
            print(f"scale {scale:g} {dtype} split={split} {metric}: {'ok' if ok else 'MISMATCH'} fallback={st['fallback_queries']} err={st['max_fast_err']:.3g} eps={st['eps_bound']:.3g}")
