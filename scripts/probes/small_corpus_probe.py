import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import vrod_amd as va
for (n, dim, dtype, metric, nq, k) in [(10000, 128, "f32", "cosine", 1, 10), (10000, 128, "f32", "cosine", 8, 10), (100000, 128, "f32", "cosine", 1, 10)]:
    ix = va.Index(dim, dtype, metric); ix.add_synthetic(1, 0, n)
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda"); osc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for _ in range(20): ix.search_synthetic_device(2, 0, nq, k, oi, osc)
    t = time.perf_counter()
    for s in range(500): ix.search_synthetic_device(2, s * nq, nq, k, oi, osc)
    sync = (time.perf_counter() - t) / 500
    bufs = [(torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda")) for _ in range(2)]
    t = time.perf_counter()
    ix.search_begin_synthetic_device(2, 0, nq, k, *bufs[0])
    for s in range(500):
        if s + 1 < 500: ix.search_begin_synthetic_device(2, (s + 1) * nq, nq, k, *bufs[(s + 1) % 2])
        ix.search_end()
    pipe = (time.perf_counter() - t) / 500
    print(f"n={n} d={dim} nq={nq}: sync {sync*1e6:.1f} us/search, pipelined {pipe*1e6:.1f} us/search")
    ix.close()
