"""ms per batch for 1..256 queries over 2M x 768 (bf16 and fp32), AUTO routing."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import vrod_amd as va
va.load()
dev = torch.device("cuda", 0)
n, d = 2_000_000, int(os.environ.get("DIM", "768"))
for dtype in ("bf16", "f32"):
    with va.Index(d, dtype, "cosine") as ix:
        ix.add_synthetic(1, 0, n)
        gb = n * d * (2 if dtype == "bf16" else 4) / 1e9
        for nq in (1, 4, 5, 8, 16, 32, 33, 64, 65, 128, 256):
            oi = torch.empty((nq, 10), dtype=torch.int64, device=dev)
            osc = torch.empty((nq, 10), dtype=torch.float32, device=dev)
            for _ in range(2):
                ix.search_synthetic_device(2, 0, nq, 10, oi, osc)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(5):
                ix.search_synthetic_device(2, i * nq, nq, 10, oi, osc)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 5 * 1e3
            st = ix.last_stats()
            print(f"{dtype} {n}x{d} nq={nq}: {ms:.3f} ms/batch (path {st['path']}, split {st['split_pass']}, fb {st['fallback_queries']}); HBM floor {gb/6.5:.3f} ms", flush=True)
