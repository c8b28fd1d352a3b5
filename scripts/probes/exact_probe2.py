"""One exact-path search and one skinny-kernel search for rocprofv3 evidence: python scripts/exact_probe2.py"""
import sys
import torch
sys.path.insert(0, ".")
import vrod_amd as va
n, dim, k = 2_000_000, 768, 10
ix = va.Index(dim, "bf16", "cosine")
ix.add_synthetic(1, 0, n)
for nq, path in ((8, va.PATH_EXACT), (32, va.PATH_MFMA), (1, va.PATH_AUTO)):
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda"); osc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    ix.set_path(path)
    for s in range(3):
        ix.search_synthetic_device(2, s * nq, nq, k, oi, osc)
    print(nq, path, ix.last_stats())
torch.cuda.synchronize()
