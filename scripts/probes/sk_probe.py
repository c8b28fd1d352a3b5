"""Scan time of batched searches of 32 / 64 queries (bf16 2M x 768): kernel time from the library's events."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import vrod_amd as va
va.load()
dev = torch.device("cuda", 0)
n, d = 2_000_000, int(os.environ.get("DIM", "768"))
with va.Index(d, "bf16", "cosine") as ix:
    ix.add_synthetic(1, 0, n)
    ix.set_profiling(2)
    for nq in (32, 64):
        oi = torch.empty((nq, 10), dtype=torch.int64, device=dev)
        osc = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        tot = sc = 0.0
        for i in range(6):
            ix.search_synthetic_device(2, i * nq, nq, 10, oi, osc)
            st = ix.last_stats()
            if i:
                tot += st["total_ms"]; sc += st["scan_ms"]
        print(f"nq={nq}: total {tot/5:.3f} ms, scans {sc/5:.3f} ms ({st['scan_launches']} launches) -> {n*d*2/1e9/(sc/5):.2f} TB/s over the scans", flush=True)
