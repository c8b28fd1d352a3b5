"""A few synchronous cfg3-shaped batches (for rocprofv3 timelines): python scripts/ab_probe.py [rows] [nq] [steps]"""
import sys
import torch
sys.path.insert(0, ".")
import vrod_amd as va
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
metric = sys.argv[5] if len(sys.argv) > 5 else "cosine"
dim = int(sys.argv[6]) if len(sys.argv) > 6 else 768
k = int(sys.argv[7]) if len(sys.argv) > 7 else 10
ix = va.Index(dim, dtype, metric)
ix.add_synthetic(1, 0, n)
ix.set_path(va.PATH_MFMA if nq > 4 else va.PATH_AUTO)
oi = torch.empty((nq, k), dtype=torch.int64, device="cuda"); osc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
for s in range(steps):
    ix.search_synthetic_device(2, s * nq, nq, k, oi, osc)
torch.cuda.synchronize()
print(ix.last_stats())
