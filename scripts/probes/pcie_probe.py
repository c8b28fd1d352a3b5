"""PCIe-inclusive rate of cfg3: vrod_search with HOST query/result buffers."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import vrod_amd as va
from oracle import oracle as O
ix = va.Index(768, "bf16", "cosine")
ix.add_synthetic(1, 0, 10_000_000)
q = O.synth_rows(2, 0, 1024, 768, threads=8)
ix.search(q, 10)
t = time.time()
for _ in range(8):
    ix.search(q, 10)
dt = (time.time() - t) / 8
print(f"host-pointer vrod_search: {dt*1e3:.3f} ms per batch of 1024 -> {1024/dt:.0f} queries/s (PCIe inclusive)")
