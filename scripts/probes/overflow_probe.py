import sys
sys.path.insert(0, ".")
import numpy as np
import vrod_amd as va
rng = np.random.default_rng(11)
for scale in (1e18, 1e19):
    raw = (rng.standard_normal((9000, 96)) * scale).astype(np.float32); raw[17] = 0
    rq = (rng.standard_normal((5, 96)) * scale).astype(np.float32); rq[0] = 0
    for path in (1, 2):
        with va.Index(96, "f32", "l2") as ix:
            ix.add(raw); ix.set_path(path)
            ids, sc = ix.search(rq, 10); st = ix.last_stats()
        print(scale, path, {k: st[k] for k in ("fallback_queries", "eps_bound", "max_fast_err", "kprime", "path")}, sc[1][:3])
