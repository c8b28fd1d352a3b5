"""Time of the exact path (path 3) per batch: nq uncertified queries over one corpus."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import vrod_amd as va
va.load()
dev = torch.device("cuda", 0)
for dtype, metric, n, d in (("bf16", "cosine", 4_000_000, 768), ("f32", "l2", 2_000_000, 768), ("f32", "cosine", 1_000_000, 1536)):
    with va.Index(d, dtype, metric) as ix:
        ix.add_synthetic(1, 0, n)
        ix.set_path(3)
        for nq in (1, 2, 4, 8, 16):
            oi = torch.empty((nq, 10), dtype=torch.int64, device=dev)
            osc = torch.empty((nq, 10), dtype=torch.float32, device=dev)
            ix.search_synthetic_device(2, 0, nq, 10, oi, osc)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                ix.search_synthetic_device(2, 0, nq, 10, oi, osc)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 3 * 1e3
            gb = n * d * (2 if dtype == "bf16" else 4) / 1e9
            print(f"{dtype} {metric} {n}x{d} nq={nq}: {ms:.2f} ms/batch, {ms/nq:.2f} ms/query, corpus {gb:.1f} GB -> one pass at 8 TB/s = {gb/8:.2f} ms", flush=True)
