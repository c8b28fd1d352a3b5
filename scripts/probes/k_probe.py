"""Cost of k' on the batched scan: python scripts/probes/k_probe.py rows nq dim dtype "k1 k2 ..."  (synchronous searches, HIP-event scan time)"""
import sys
import time
import torch
sys.path.insert(0, ".")
import vrod_amd as va
n, nq, dim, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ks = [int(x) for x in sys.argv[5].split()]
ix = va.Index(dim, dtype, "cosine")
ix.add_synthetic(1, 0, n)
ix.set_path(va.PATH_MFMA)
ix.set_profiling(True)
for k in ks:
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda"); osc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for s in range(2):
        ix.search_synthetic_device(2, s * nq, nq, k, oi, osc)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sc = 0.0; reps = 4
    for s in range(reps):
        ix.search_synthetic_device(2, (2 + s) * nq, nq, k, oi, osc)
        st = ix.last_stats(); sc += st["scan_ms"]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    fl = st["scan_flops"] * (3 if st["split_pass"] else 1)
    print(f"k={k} kprime={st['kprime']} launches={st['scan_launches']} split={st['split_pass']} wall={wall:.3f} ms scan={sc / reps:.3f} ms  {fl / (sc / reps * 1e-3) / 1e12:.0f} TF executed  fallback={st['fallback_queries']}", flush=True)
