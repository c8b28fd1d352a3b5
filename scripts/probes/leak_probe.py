import sys
sys.path.insert(0, ".")
import numpy as np, torch
import vrod_amd as va
rng = np.random.default_rng(1)
raw = rng.standard_normal((20000, 128)).astype(np.float32)
rq = rng.standard_normal((40, 128)).astype(np.float32)
free0 = None
for it in range(120):
    with va.Index(128, ["f32", "bf16"][it % 2], ["cosine", "l2"][(it // 2) % 2]) as ix:
        ix.add(raw)
        ix.search(rq, 10)
        ix.search(rq[:2], 10)
        if it % 3 == 0:
            with va.Index(128, "f32", "cosine", devices=[0, 0]) as mx:
                mx.add(raw); mx.search(rq[:3], 5)
    if it in (10, 60, 119):
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        print(f"iteration {it}: free {free/2**20:.0f} MiB")
