"""Where should AUTO switch from the stream path to the MFMA path? (dev probe)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import vrod_amd as va
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for dtype in ("bf16", "f32"):
    ix = va.Index(768, dtype, "cosine"); ix.add_synthetic(1, 0, n); ix.set_profiling(True)
    for nq in (1, 2, 3, 4, 8, 16):
        oi = torch.empty((nq, 10), dtype=torch.int64, device="cuda"); osc = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
        row = []
        for path in (1, 2):
            ix.set_path(path)
            ix.search_synthetic_device(2, 0, nq, 10, oi, osc)
            ts = []
            for s in range(5):
                t = time.time(); ix.search_synthetic_device(2, s * nq, nq, 10, oi, osc); ts.append(time.time() - t)
            row.append(np.median(ts) * 1e3)
        print(f"{dtype} n={n} nq={nq:3d}  stream {row[0]:7.3f} ms   mfma {row[1]:7.3f} ms", flush=True)
    ix.close()
