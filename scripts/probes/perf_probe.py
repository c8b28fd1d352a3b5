"""Quick perf probe of the two headline configs (not the bench contract; see bench.py)."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
import vrod_amd as va

def probe(name, n, dim, dtype, metric, nq, k, steps=5, path=0):
    ix = va.Index(dim, dtype, metric)
    t = time.time(); ix.add_synthetic(1, 0, n); torch.cuda.synchronize(); tg = time.time() - t
    ix.set_profiling(2); ix.set_path(path)
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda"); osc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    ix.search_synthetic_device(2, 0, nq, k, oi, osc)  # warmup
    res = []
    for s in range(steps):
        t = time.time(); ix.search_synthetic_device(2, s * nq, nq, k, oi, osc); dt = time.time() - t
        st = ix.last_stats(); st["wall_ms"] = dt * 1e3; res.append(st)
    st = res[-1]
    scan = np.median([r["scan_ms"] for r in res]); tot = np.median([r["total_ms"] for r in res]); wall = np.median([r["wall_ms"] for r in res])
    print(json.dumps({"cfg": name, "gen_s": round(tg, 2), "scan_ms": round(float(scan), 3), "total_ms": round(float(tot), 3), "wall_ms": round(float(wall), 3),
                      "GBps": round(st["scan_bytes"] / scan / 1e6, 1), "TFLOPs": round(st["scan_flops"] / scan / 1e9, 1), "launches": st["scan_launches"],
                      "fallback": sum(r["fallback_queries"] for r in res), "kprime": st["kprime"], "err": st["max_fast_err"], "eps": st["eps_bound"],
                      "qps": round(nq / wall * 1e3, 1)}), flush=True)
    ix.close()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "cfg2"):
    probe("cfg2 1Mx768 f32 L2 Q=1 k=100", 1_000_000, 768, "f32", "l2", 1, 100, steps=10)
    probe("cfg2b 1Mx768 bf16 cos Q=1 k=10", 1_000_000, 768, "bf16", "cosine", 1, 10, steps=10)
    probe("cfg2c 1Mx768 f32 cos Q=8 k=10", 1_000_000, 768, "f32", "cosine", 8, 10, steps=5)
if which in ("all", "cfg3s"):
    probe("cfg3-small 1Mx768 bf16 cos Q=1024 k=10", 1_000_000, 768, "bf16", "cosine", 1024, 10, steps=3)
if which in ("all", "cfg3"):
    probe("cfg3 10Mx768 bf16 cos Q=1024 k=10", 10_000_000, 768, "bf16", "cosine", 1024, 10, steps=3)
if which in ("l2bf16",):
    probe("4Mx768 bf16 L2 Q=1024 k=10", 4_000_000, 768, "bf16", "l2", 1024, 10, steps=3)
if which in ("cfg5",):
    probe("cfg5 10Mx1536 f32 cos Q=256 k=1000", 10_000_000, 1536, "f32", "cosine", 256, 1000, steps=2)
