"""Small corpora: stream scan vs MFMA (skinny) path for 5-32 queries, sync us per search."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import vrod_amd as va
va.load()
for dtype in ("f32", "bf16"):
    for n in (10_000, 100_000, 400_000, 1_000_000):
        for d in (128, 768):
            with va.Index(d, dtype, "cosine") as ix:
                ix.add_synthetic(1, 0, n)
                for nq in [int(x) for x in os.environ.get("NQS", "5,8,16,32").split(",")]:
                    oi = torch.empty((nq, 10), dtype=torch.int64, device="cuda"); osc = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
                    res = []
                    for path in (1, 2):
                        ix.set_path(path)
                        for _ in range(5): ix.search_synthetic_device(2, 0, nq, 10, oi, osc)
                        torch.cuda.synchronize()
                        t = time.perf_counter()
                        for s in range(40): ix.search_synthetic_device(2, s * nq, nq, 10, oi, osc)
                        torch.cuda.synchronize()
                        res.append((time.perf_counter() - t) / 40 * 1e6)
                    ix.set_path(0)
                    ix.search_synthetic_device(2, 0, nq, 10, oi, osc)
                    print(f"{dtype} n={n} d={d} nq={nq}: stream {res[0]:.0f} us, mfma {res[1]:.0f} us, auto -> path {ix.last_stats()['path']}", flush=True)
