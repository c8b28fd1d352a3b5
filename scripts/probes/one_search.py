"""A few searches of one shape (for rocprofv3 timelines): NQ, N, DIM, DTYPE from the environment."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import vrod_amd as va
va.load()
dev = torch.device("cuda", 0)
nq, n, d = int(os.environ.get("NQ", "32")), int(os.environ.get("N", "2000000")), int(os.environ.get("DIM", "768"))
with va.Index(d, os.environ.get("DTYPE", "bf16"), os.environ.get("METRIC", "cosine")) as ix:
    ix.add_synthetic(1, 0, n)
    oi = torch.empty((nq, 10), dtype=torch.int64, device=dev)
    osc = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    for i in range(4):
        ix.search_synthetic_device(2, i * nq, nq, 10, oi, osc)
    torch.cuda.synchronize()
    print(ix.last_stats())
